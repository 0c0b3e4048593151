# last call of the round: Grover and the 1024-gate configs[1] circuit in the IEEE-equal mode with the final scheduler
# (first-come positions in tile = 1, pass-minimising gate order inside the segments)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03w
QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 25 python tools/bench_tile.py 30 3 grover,c2x4 1 > gpurun_out/r03w/final.jsonl 2> gpurun_out/r03w/final.err
cut -c1-160 gpurun_out/r03w/final.jsonl
