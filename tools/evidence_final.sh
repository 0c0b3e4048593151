# The round's closing pass on the GPU box, after the last change to csrc/qip_kernels.h: PMC + rocprofv3 passes first (so that the
# bench line's roofline.traffic comes from the very tree it runs on), then the contract bench line and the per-op tables;
# r5: + PMC passes over every run-time-compiled segment of the wide legs (configs[1] relabelled, Clifford+T relabelled, 1e-12 mode).
#   gpurun -- 'bash tools/evidence_final.sh r05'       results under gpurun_out/<tag>/ and gpurun_out/<tag>f/
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
F=$R/gpurun_out/${TAG}f
mkdir -p $O $F
cd $R
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
python tools/summarize_rocprof.py $TAG $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_prof.json $O/prof_stats_extras > $O/summarize.log 2>&1
S=$(date +%s)
timeout 560 python bench.py --steps 20 --warmup 5 > $F/bench_n1.json 2> $F/bench_n1.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee $F/bench_wall.txt
# a second process of the same command: every segment comes from the disk cache (extras.jit of that line)
S=$(date +%s)
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $F/bench_n1_second_process.json 2> $F/bench_n1_second_process.err
echo "second bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $F/bench_wall.txt
timeout 200 python tools/bench_ops.py 30 all > $O/ops_table.md 2> $O/ops_table.err
timeout 200 python tools/bench_ops.py 30 all f32 > $O/ops_table_f32.md 2> $O/ops_table_f32.err
# PMC over the run-time-compiled wide segments (counters only + kernel trace: the combination gpurun allows)
cd /tmp && export TMPDIR=/tmp
for leg in "c2 1 1" "c4 1 1" "c2 2 0"; do
  set -- $leg
  T=jit_$1_tile$2_relabel$3
  for C in FETCH_SIZE WRITE_SIZE; do
    QIP_TILE_JIT=1 QIP_TILE_WIDE=1 QIP_TILE_RELABEL=$3 QIP_TILE_FMA=$(( $2 - 1 )) QIP_TILE_MERGE=$(( $2 - 1 )) rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${T}_$C -o $T --output-format csv -- \
      python $R/tools/bench_tile.py 30 1 $1 $2 > $O/pmc_${T}_$C.log 2>&1
  done
  python $R/tools/pmc_jit_segments.py $O/pmc_${T}_FETCH_SIZE $O/pmc_${T}_WRITE_SIZE "$1, tile = $2, wide tiles, relabel = $3" >> $O/jit_segments_pmc.md 2>> $O/jit_segments_pmc.err
done
cd $R
# two ranks on the one GPU (host-staged transport): plumbing + parity + the overlapped exchange's counters, not a throughput figure
QIP_BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --n-local 24 --steps 3 --warmup 1 --dist-overlap 4 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
timeout 900 python -m pytest tests -m gpu -q --durations=6 -k "sharded_virtual or bench_multi_rank or programs_compile" > $O/gpu_tests_subset.txt 2>&1
tail -8 $O/gpu_tests_subset.txt
tail -c 400 $F/bench_n1.json
