cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "dense4 or dense_k or big_k or every_kernel or special" --durations=5 > $O/pytest_k4.txt 2>&1
tail -8 $O/pytest_k4.txt
timeout 300 python tools/bench_ops.py 30 "dense k=4" > $O/ops_k4.md 2>&1
QIP_K4_DIRECT=1 timeout 300 python tools/bench_ops.py 30 "dense k=4" > $O/ops_k4_direct.md 2>&1
timeout 300 python tools/bench_ops.py 30 "dense k=4" f32 > $O/ops_k4_f32.md 2>&1
grep "dense k=4" $O/ops_k4.md; echo direct; grep "dense k=4" $O/ops_k4_direct.md; echo f32; grep "dense k=4" $O/ops_k4_f32.md
