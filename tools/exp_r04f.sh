R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=5 -k "wide_tiles and complex128" > $O/tests.txt 2>&1; tail -6 $O/tests.txt
for W in 1; do
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft,grover,groverk3 1 >> $O/tile.jsonl 2>> $O/err.txt
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4 1 >> $O/tile.jsonl 2>> $O/err.txt
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft 2 >> $O/tile.jsonl 2>> $O/err.txt
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4 2 >> $O/tile.jsonl 2>> $O/err.txt
done
python - <<PY
import json
for l in open("$O/tile.jsonl"):
    d=json.loads(l); print(d["circuit"],"tile",d["tile"],"relabel",d["relabel"],"fma",d["fma"],d["sweeps"],d["ms"])
PY
tail -3 $O/err.txt
