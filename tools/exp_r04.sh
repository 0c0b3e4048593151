# Round-4 experiments, one GPU call each (run through gpurun: `gpurun -- 'bash tools/exp_r04.sh <step>'`); results under gpurun_out/r04x/.
#   rows     split rows (tile_row_split = 11) against contiguous rows (5): tile sweeps and one-op sweeps   -> profiles/r04_tile_rows.md §3
#   probe    the address-bit study: tools/tune_tile scan / probe (stdin: tools/probe_in.txt) / v13         -> profiles/r04_tile_rows.md §1, 2, 4
#   sparse   k_sparse_tile: parity tests + the sparse rows of the ops table with the out-of-place gather beside it
#   wide     wide tiles against the 11-bit sweeps on the benchmark circuits                               -> profiles/r04_wide_tiles.md
STEP=${1:-wide}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
case $STEP in
rows)
  for RS in 5 11; do
    QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4,qft,grover,groverk3 1 >> $O/tile_rs$RS.jsonl
    QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4 1 >> $O/tile_rs$RS.jsonl
    QIP_TILE_ROW_SPLIT=$RS QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 timeout 300 python tools/bench_tile.py 30 5 c2,c4,qft 2 >> $O/tile_rs$RS.jsonl
    QIP_TILE_ROW_SPLIT=$RS timeout 300 python tools/bench_ops.py 30 "dense" > $O/ops_dense_rs$RS.md
  done ;;
probe)
  timeout 250 tools/tune_tile 30 5 scan > $O/scan.txt 2>&1
  timeout 250 tools/tune_tile 30 5 probe < tools/probe_in.txt > $O/probe.txt 2>&1
  timeout 250 tools/tune_tile 30 5 v13 > $O/v13.txt 2>&1 ;;
wide)
  for W in 0 1; do
    QIP_TILE_WIDE=$W QIP_TILE_JIT=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft,grover,groverk3 1 >> $O/tile_wide.jsonl
    QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4 1 >> $O/tile_wide.jsonl
    QIP_TILE_WIDE=$W QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 timeout 400 python tools/bench_tile.py 30 5 c2,c4,qft 2 >> $O/tile_wide.jsonl
  done ;;
sparse)
  timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "sparse or full_size_oracle_windows" > $O/sparse_tests.txt 2>&1; tail -5 $O/sparse_tests.txt
  timeout 300 python tools/bench_ops.py 30 sparse > $O/ops_sparse.md 2> $O/ops_sparse.err; cat $O/ops_sparse.md
  timeout 300 python tools/bench_ops.py 30 sparse f32 > $O/ops_sparse_f32.md 2>> $O/ops_sparse.err; cat $O/ops_sparse_f32.md ;;
esac
