# round-3 check D: the new parity machinery on the GPU (two new tests + bench parity block at a small size)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "side_by_side or every_timed_leg" > gpurun_out/r03d/pytest.txt 2>&1
tail -30 gpurun_out/r03d/pytest.txt
timeout 600 python bench.py --n-local 26 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03d/bench_n26.json 2> gpurun_out/r03d/bench_n26.err
tail -5 gpurun_out/r03d/bench_n26.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d/bench_n26.json').read().strip().split('\n')[-1])
p=d['parity']
print({k:v for k,v in p.items() if k!='legs'})
for k,v in p['legs'].items(): print(k, {a:b for a,b in v.items() if a not in ('options','product_state_marginals')})
PY
