# r4: one-op sweeps for controlled dense gates, k = 4 on the matrix cores by default, k = 9 / 10
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=6 -k "one_op_tile or dense_big_k or dense_k_qubit or dense4_on or (full_size_oracle_windows and 28)" > $O/tests.txt 2>&1; tail -12 $O/tests.txt
timeout 600 python tools/bench_ops.py 30 "dense" > $O/ops_dense.md 2> $O/err.txt; tail -3 $O/err.txt
grep -E "controlled|k=4|k=9|k=10|k=8" $O/ops_dense.md
