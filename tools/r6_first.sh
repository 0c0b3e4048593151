#!/bin/bash
# round 6, first GPU pass: the new program / one-shot tests, then the default bench run
mkdir -p gpurun_out/r06a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export QIP_BENCH_DETAIL=gpurun_out/r06a/bench_detail.json
timeout 900 python -m pytest tests/test_gpu_f4_tiles.py -x -q -k "program or one_shot" > gpurun_out/r06a/tests_programs.txt 2>&1
echo "rc=$?" >> gpurun_out/r06a/tests_programs.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06a/bench_stdout.txt 2> gpurun_out/r06a/bench_stderr.txt
echo "rc=$?" >> gpurun_out/r06a/bench_stderr.txt
tail -c 600 gpurun_out/r06a/tests_programs.txt
tail -n 3 gpurun_out/r06a/bench_stdout.txt | cut -c1-1500
tail -n 12 gpurun_out/r06a/bench_stderr.txt
