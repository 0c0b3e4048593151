#!/bin/bash
# the round's last GPU call: the full GPU suite, then the bench run whose line is committed (PMC traffic of the same kernels: not stale)
out=gpurun_out/${1:-final}
mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $out/gpu_tests.txt 2>&1
echo "rc=$?" >> $out/gpu_tests.txt
tail -n 22 $out/gpu_tests.txt
export QIP_BENCH_DETAIL=$out/bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $out/bench_n1.json 2> $out/bench_n1.err
tail -n 1 $out/bench_n1.json | cut -c1-300
grep -v amdgpu.ids $out/bench_n1.err | tail -n 6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
