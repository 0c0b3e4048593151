# XCD-aware block -> tile order in the run-time-compiled segments (global option tile_remap = 4) against the plain order:
# the HBM-bound tile mode (tile = 2, fused multiply-adds, merged diagonal runs, relabelled) on configs[1] and QFT.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03x
mkdir -p $O
export QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1
for R in 0 4; do
  QIP_TILE_REMAP=$R timeout 170 python tools/bench_tile.py 30 5 c2,qft 2 > $O/remap$R.jsonl 2> $O/remap$R.err
done
cat $O/remap0.jsonl $O/remap4.jsonl
