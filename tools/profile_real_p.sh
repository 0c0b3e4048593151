# rocprofv3 evidence of the real-P kernels (run on the GPU box through gpurun): kernel stats and the two separate PMC passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950) of `tools/bench_real_p.py --profile`: n = 28, f64 and f32, four shapes
# x three calls.   usage: bash tools/profile_real_p.sh r06   -> gpurun_out/<tag>_realp/ ; summary: tools/summarize_real_p.py
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_realp
mkdir -p $O
W="python $R/tools/bench_real_p.py --profile"
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $TAG --output-format csv -- $W > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o $TAG --output-format csv -- $W > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o $TAG --output-format csv -- $W > $O/write.log 2>&1
ls $O/prof_stats $O/prof_fetch $O/prof_write
