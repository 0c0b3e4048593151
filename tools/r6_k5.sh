#!/bin/bash
out=gpurun_out/${1:-r06k}
mkdir -p $out
for p in 8 16 32 64 256; do
  echo "== QIP_K5_PIPE=$p"
  QIP_K5_PIPE=$p timeout 300 python -m pytest tests -m gpu -x -q -k "dense5_on or dense4_on" 2>&1 | tail -n 1
  QIP_K5_PIPE=$p timeout 300 python tools/bench_ops.py 30 "dense k=5" > $out/ops_dense_f64_pipe$p.md 2>&1
  QIP_K5_PIPE=$p timeout 300 python tools/bench_ops.py 30 "dense k=5" f32 > $out/ops_dense_f32_pipe$p.md 2>&1
  grep "k=5" $out/ops_dense_f64_pipe$p.md $out/ops_dense_f32_pipe$p.md | grep -v literal | cut -c1-160
done
timeout 300 python tools/bench_ops.py 30 "dense k=4" > $out/ops_dense4_f64.md 2>&1
grep "k=4" $out/ops_dense4_f64.md
