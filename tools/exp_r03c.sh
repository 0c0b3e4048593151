# round-3 experiment C: tile-bit padding / wave-bit roles / block remap on real circuits (run-time-compiled segments, tile = 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
out=gpurun_out/r03c/bench_tile_tune.jsonl
run() { env "$@" QIP_TILE_JIT=1 timeout 300 python tools/bench_tile.py 30 3 c2,c4,grover 1 >> $out 2>&1; }
run QIP_TILE_PAD_FROM=6
run QIP_TILE_PAD_FROM=11
run QIP_TILE_PAD_FROM=11 QIP_TILE_WAVE_RULE=1
run QIP_TILE_PAD_FROM=11 QIP_TILE_WAVE_RULE=2
run QIP_TILE_PAD_FROM=11 QIP_TILE_REMAP=2
run QIP_TILE_PAD_FROM=11 QIP_TILE_REMAP=3
run QIP_TILE_PAD_FROM=11 QIP_TILE_WAVE_RULE=1 QIP_TILE_REMAP=3
run QIP_TILE_PAD_FROM=11 QIP_TILE_RELABEL=1
run QIP_TILE_PAD_FROM=11 QIP_TILE_RELABEL=1 QIP_TILE_WAVE_RULE=1
cat $out
