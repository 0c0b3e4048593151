R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=5 -k "wide_tiles" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for W in 0 1; do
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft,grover,groverk3 1 >> $O/tile.jsonl 2>> $O/err.txt
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4 1 >> $O/tile.jsonl 2>> $O/err.txt
QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft 2 >> $O/tile.jsonl 2>> $O/err.txt
done
python - <<PY
import json
rows={}
for l in open("$O/tile.jsonl"):
    d=json.loads(l); rows.setdefault((d["circuit"],d["tile"],d["relabel"],d["fma"]),{})[d["wide"]]=(d["sweeps"],d["ms"])
for k,v in rows.items(): print(k, v.get("0"), v.get("1"))
PY
tail -3 $O/err.txt
