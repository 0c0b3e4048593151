R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=5 -k "measure or norm or grover or dense3 or (tile and jit)" > $O/tests.txt 2>&1; tail -9 $O/tests.txt
QIP_TILE_JIT=1 timeout 300 python tools/bench_tile.py 30 5 groverk3,grover 1 > $O/tile_grover.jsonl 2>> $O/err.txt
QIP_TILE_JIT=0 timeout 300 python tools/bench_tile.py 30 5 groverk3,grover 1 >> $O/tile_grover.jsonl 2>> $O/err.txt
python - <<PY
import json
for l in open("$O/tile_grover.jsonl"):
    d=json.loads(l); print(d["circuit"], "jit", d["jit"], d["sweeps"], d["ms"])
PY
timeout 300 python tools/bench_ops.py 30 zzz f32 > $O/ops_f32_red.md 2>> $O/err.txt; grep -E "norm_sqr|measure" $O/ops_f32_red.md
tail -3 $O/err.txt
