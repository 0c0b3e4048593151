#!/bin/bash
out=gpurun_out/${1:-r06w}
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "sparse" 2>&1 | tail -n 3
timeout 300 python tools/bench_ops.py 30 "sparse" > $out/ops_sparse_f64.md 2>&1
timeout 300 python tools/bench_ops.py 30 "sparse" f32 > $out/ops_sparse_f32.md 2>&1
grep "sparse" $out/ops_sparse_f64.md $out/ops_sparse_f32.md | cut -c1-200
