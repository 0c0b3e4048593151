#!/bin/bash
out=gpurun_out/${1:-r06g}
mkdir -p $out
for p in 0 1; do
  echo "== QIP_PERM_F32_SPLIT=$p"
  QIP_PERM_F32_SPLIT=$p timeout 300 python tools/bench_permute.py 30 f32 > $out/permute_f32_split$p.md 2>&1
  QIP_PERM_F32_SPLIT=$p timeout 300 python -m pytest tests -m gpu -x -q -k "permut" 2>&1 | tail -n 1
  cat $out/permute_f32_split$p.md | grep "^| [a-z]" | cut -c1-90
done
