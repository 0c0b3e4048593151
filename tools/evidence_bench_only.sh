R=${GRAFT_REPO_ROOT:-/root/repo}; F=$R/gpurun_out/r05h; mkdir -p $F; cd $R
export QIP_HIP_CACHE_DIR=/tmp/qip_hip_cache_bench
S=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 > $F/bench_n1.json 2> $F/bench_n1.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee $F/bench_wall.txt
S=$(date +%s)
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $F/bench_n1_second_process.json 2> $F/bench_n1_second_process.err
echo "second bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $F/bench_wall.txt
