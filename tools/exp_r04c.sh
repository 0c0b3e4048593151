# r4: fold of the remap's gather into the preceding tile sweep, packed f32 reductions, k = 10 in two phases
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=6 -k "sharded_virtual or measure or norm or dense_big_k" > $O/tests.txt 2>&1; tail -12 $O/tests.txt
timeout 300 python tools/bench_ops.py 30 "k=10" > $O/ops_k10.md 2>> $O/err.txt; grep "k=" $O/ops_k10.md | head -3
grep -E "norm_sqr|measure" $O/ops_k10.md
timeout 300 python tools/bench_ops.py 30 "zzz" all f32 > $O/ops_f32_red.md 2>> $O/err.txt; grep -E "norm_sqr|measure" $O/ops_f32_red.md
tail -3 $O/err.txt
