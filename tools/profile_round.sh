# rocprofv3 evidence for one round (run on the GPU box through gpurun): kernel stats of the headline bench, the same
# with every extras leg, and the two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
# usage: bash tools/profile_round.sh r02        results under gpurun_out/<tag>/prof_*
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
sha256sum $R/rustqip_amd/csrc/qip_kernels.h | cut -c1-16 > $O/kernels_sha16.txt   # which kernels these passes describe
B="python $R/bench.py --no-cpu-baseline --no-parity"
H="python $R/bench.py --headline-only"
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $TAG --output-format csv -- $H --steps 3 --warmup 1 > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o $TAG --output-format csv -- $H --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o $TAG --output-format csv -- $H --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --stats -d $O/prof_stats_extras -o $TAG --output-format csv -- $B --steps 2 --warmup 1 > $O/bench_prof_extras.json 2> $O/bench_prof_extras.err
ls $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_stats_extras
