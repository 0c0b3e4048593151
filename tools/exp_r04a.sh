# r4 A/B: split rows (tile_row_split = 11) against contiguous rows (5) on the tile sweeps and the one-op sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "(full_size_oracle_windows and 28) or tile_relabel_is_bit or one_op_tile or k4_tile" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for RS in 5 11; do
  QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4,qft,grover,groverk3 1 >> $O/tile_rs$RS.jsonl 2>> $O/err.txt
  QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4 1 >> $O/tile_rs$RS.jsonl 2>> $O/err.txt
  QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4,qft 2 >> $O/tile_rs$RS.jsonl 2>> $O/err.txt
  QIP_TILE_ROW_SPLIT=$RS timeout 300 python tools/bench_ops.py 30 "dense" > $O/ops_dense_rs$RS.md 2>> $O/err.txt
  QIP_TILE_ROW_SPLIT=$RS timeout 300 python tools/bench_ops.py 30 "H, target" >> $O/ops_dense_rs$RS.md 2>> $O/err.txt
  QIP_TILE_ROW_SPLIT=$RS timeout 300 python tools/bench_ops.py 30 "Swap" >> $O/ops_dense_rs$RS.md 2>> $O/err.txt
done
tail -5 $O/err.txt
