cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -x -k "sparse or soft_measure or persists or sharded or dist or virtual" --durations=8 > gpurun_out/r03h/pytest.txt 2>&1
tail -22 gpurun_out/r03h/pytest.txt
timeout 300 python tools/bench_ops.py 30 "sparse" > gpurun_out/r03h/ops_sparse.md 2>&1
cat gpurun_out/r03h/ops_sparse.md | grep -v "reduction\|norm"
