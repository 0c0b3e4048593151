# The round's closing pass on the GPU box, after the last change to csrc/qip_kernels.h: PMC + rocprofv3 passes first (so that the
# bench line's roofline.traffic comes from the very tree it runs on), then the contract bench line and the per-op tables.
#   gpurun -- 'bash tools/evidence_final.sh r04'       results under gpurun_out/<tag>/ and gpurun_out/<tag>f/
TAG=${1:-r04}
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
timeout 200 python tools/bench_ops.py 30 all > $O/ops_table.md 2> $O/ops_table.err
timeout 200 python tools/bench_ops.py 30 all f32 > $O/ops_table_f32.md 2> $O/ops_table_f32.err
tail -c 400 $F/bench_n1.json
