#!/bin/bash
# round-end extras: the headline at n = 33 on ONE GPU (128 GiB: the per-GPU-times-eight size of BASELINE configs[4]), two ranks x 2^29 on one GPU
# through the host-staged test transport, and the launch-bound regime (n = 8..24)
out=gpurun_out/${1:-r06q}
mkdir -p $out
QIP_BENCH_DETAIL=$out/bench_detail_n33.json timeout 900 python bench.py --headline-only --n-local 33 --gates 64 --steps 2 --warmup 1 > $out/bench_n33.json 2> $out/bench_n33.err
tail -n 1 $out/bench_n33.json | cut -c1-400
QIP_BENCH_DETAIL=$out/bench_detail_2ranks_n29.json QIP_BENCH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --n-local 29 --steps 2 --warmup 1 --headline-only > $out/bench_2ranks_one_gpu_n29.json 2> $out/bench_2ranks_one_gpu_n29.err
tail -n 1 $out/bench_2ranks_one_gpu_n29.json | cut -c1-400
timeout 600 python tools/bench_small_n.py > $out/small_n_launch_bound.md 2> $out/small_n.err
tail -n 8 $out/small_n_launch_bound.md | cut -c1-200
