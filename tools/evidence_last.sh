# The last GPU call of the round: the full GPU suite on a fresh box (cold caches: its wall time is what the driver will see), then the
# contract bench line from the final bench.py with a cache directory of its own (cold), and a second process on the same directory (warm).
#   gpurun -- 'bash tools/evidence_last.sh r05'
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$R/gpurun_out/${TAG}g
mkdir -p $F
cd $R
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > $F/gpu_tests.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $F/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $F/gpu_tests.txt 2>&1
export QIP_HIP_CACHE_DIR=/tmp/qip_hip_cache_bench
S=$(date +%s)
timeout 560 python bench.py --steps 20 --warmup 5 > $F/bench_n1.json 2> $F/bench_n1.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee $F/bench_wall.txt
S=$(date +%s)
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $F/bench_n1_second_process.json 2> $F/bench_n1_second_process.err
echo "second bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $F/bench_wall.txt
tail -c 300 $F/bench_n1.json
