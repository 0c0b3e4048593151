#!/bin/bash
# round 6: the split-row bit permutation — parity tests that run it, then tools/bench_permute.py (f64, f32) and a rocprofv3 kernel summary
out=gpurun_out/${1:-r06d}
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "permut or swap or relabel or settle or remap or qft" > $out/tests_permute.txt 2>&1
echo "rc=$?" >> $out/tests_permute.txt
timeout 300 python tools/bench_permute.py 30 > $out/permute_f64.md 2>&1
timeout 300 python tools/bench_permute.py 30 f32 > $out/permute_f32.md 2>&1
timeout 300 python tools/bench_permute.py 31 f32 > $out/permute_f32_n31.md 2>&1
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o permute -- python $GRAFT_REPO_ROOT/tools/bench_permute.py 30 ) > $out/rocprof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/permute_kernel_stats.csv \;
find $out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
tail -n 5 $out/tests_permute.txt
cat $out/permute_f64.md $out/permute_f32.md
head -8 $out/permute_kernel_stats.csv | cut -c1-200
