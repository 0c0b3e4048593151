cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r03g/pytest_gpu.txt 2>&1
tail -25 gpurun_out/r03g/pytest_gpu.txt
timeout 600 python tools/bench_ops.py 30 all > gpurun_out/r03g/ops_table.md 2> gpurun_out/r03g/ops_table.err
cat gpurun_out/r03g/ops_table.md; tail -3 gpurun_out/r03g/ops_table.err
