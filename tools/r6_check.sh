#!/bin/bash
# the final tree once more: the full GPU suite, then the default-shaped bench run with the committed PMC traffic of the same kernels
out=gpurun_out/${1:-r06y}
mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $out/gpu_tests.txt 2>&1
echo "rc=$?" >> $out/gpu_tests.txt
tail -n 22 $out/gpu_tests.txt
export QIP_BENCH_DETAIL=$out/bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $out/bench_n1.json 2> $out/bench_n1.err
tail -n 1 $out/bench_n1.json | cut -c1-300
grep -v amdgpu.ids $out/bench_n1.err | tail -n 6
( time timeout 600 python bench.py ) > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-200; grep real $out/bench_default.err
