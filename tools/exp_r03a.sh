# round-3 experiment A: tile-sweep skeleton variants + fused multiply-adds in run-time-compiled tile = 2 segments
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 300 tools/tune_tile 30 5 > gpurun_out/r03a/tune_tile.txt 2>&1
for fma in 0 1; do
  QIP_TILE_JIT=1 QIP_TILE_FMA=$fma timeout 300 python tools/bench_tile.py 30 3 c2,c4,qft,grover 2 >> gpurun_out/r03a/bench_tile_fma.jsonl 2>&1
done
QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 timeout 300 python tools/bench_tile.py 30 3 c2,c4 2 >> gpurun_out/r03a/bench_tile_fma.jsonl 2>&1
QIP_TILE_JIT=1 timeout 300 python tools/bench_tile.py 30 3 qft,c2 1 >> gpurun_out/r03a/bench_tile_fma.jsonl 2>&1
tail -n 100 gpurun_out/r03a/tune_tile.txt
cat gpurun_out/r03a/bench_tile_fma.jsonl
