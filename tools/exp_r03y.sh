# positions of a segment claimed first-come (tile_sched = 0) against the search over the claiming rules (1, the default):
# configs[1], its 1024-gate version and Clifford+T at n = 30, IEEE-equal mode and the 1e-12 mode, relabelled, run-time-compiled
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03y
mkdir -p $O
export QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1
for R in 0 1; do
  QIP_TILE_SCHED=$R timeout 100 python tools/bench_tile.py 30 5 c2,c4,c2x4,grover 1,2 > $O/sched$R.jsonl 2> $O/sched$R.err
done
python - <<'PY'
import json
for r in (0, 1):
    for l in open(f"gpurun_out/r03y/sched{r}.jsonl"):
        d = json.loads(l)
        print(r, d["circuit"], d["tile"], d["sweeps"], d["ms"], d["norm"])
PY
