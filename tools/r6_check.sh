#!/bin/bash
out=gpurun_out/${1:-r06n}
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "dense5_on or dense4_on or dense_big or fuse or special" 2>&1 | tail -n 6
timeout 300 python tools/bench_ops.py 30 "dense k=" > $out/ops_dense_f64.md 2>&1
timeout 300 python tools/bench_ops.py 30 "dense k=" f32 > $out/ops_dense_f32.md 2>&1
grep "dense k=[5-9]\|dense k=10" $out/ops_dense_f64.md $out/ops_dense_f32.md | grep -v literal | cut -c1-170
