cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
for v in 0 1 2; do
  echo "## single_via_tile = $v" >> gpurun_out/r03f/ops_single_via_tile.md
  QIP_SINGLE_VIA_TILE=$v timeout 300 python tools/bench_ops.py 30 "dense k=2" >> gpurun_out/r03f/ops_single_via_tile.md 2>&1
  QIP_SINGLE_VIA_TILE=$v timeout 300 python tools/bench_ops.py 30 "dense k=3" >> gpurun_out/r03f/ops_single_via_tile.md 2>&1
  QIP_SINGLE_VIA_TILE=$v timeout 300 python tools/bench_ops.py 30 "Swap(" >> gpurun_out/r03f/ops_single_via_tile.md 2>&1
done
grep -v "norm_sqr\|measure\|reduction\|^$\|norm after" gpurun_out/r03f/ops_single_via_tile.md
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "dense or swap or Swap" 2>&1 | tail -5
