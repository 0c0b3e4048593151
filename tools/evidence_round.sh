# Everything the round's profiles/ files come from, in one GPU call (run through gpurun):
#   bash tools/evidence_round.sh r04 [quick]
# results under gpurun_out/<tag>/ ; tools/summarize_rocprof.py and the copy into profiles/ run afterwards on the build box.
#   bash tools/evidence_round.sh r06 final    the round's LAST GPU call: the full GPU suite, the bench run whose line is committed, smoke
#   bash tools/evidence_round.sh r06 extras   the headline at n = 33 on ONE GPU (128 GiB), two ranks x 2^29 on one GPU (host-staged
#                                             transport), the launch-bound regime (n = 8..24), the real-P table
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" = "final" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $O/gpu_tests.txt 2>&1
  echo "rc=$?" >> $O/gpu_tests.txt
  tail -n 22 $O/gpu_tests.txt
  export QIP_BENCH_DETAIL=$O/bench_detail.json
  ( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err
  tail -n 1 $O/bench_n1.json | cut -c1-300
  grep -v amdgpu.ids $O/bench_n1.err | tail -n 6
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
  exit 0
fi
if [ "$2" = "extras" ]; then
  QIP_BENCH_DETAIL=$O/bench_detail_n33.json timeout 900 python bench.py --headline-only --n-local 33 --gates 64 --steps 2 --warmup 1 > $O/bench_n33.json 2> $O/bench_n33.err
  tail -n 1 $O/bench_n33.json | cut -c1-400
  QIP_BENCH_DETAIL=$O/bench_detail_2ranks_n29.json QIP_BENCH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --n-local 29 --steps 2 --warmup 1 --headline-only > $O/bench_2ranks_one_gpu_n29.json 2> $O/bench_2ranks_one_gpu_n29.err
  tail -n 1 $O/bench_2ranks_one_gpu_n29.json | cut -c1-400
  timeout 600 python tools/bench_small_n.py > $O/small_n_launch_bound.md 2> $O/small_n.err
  tail -n 8 $O/small_n_launch_bound.md | cut -c1-200
  timeout 300 python tools/bench_real_p.py > $O/real_p.md 2> $O/real_p.err
  tail -n 6 $O/real_p.md | cut -c1-200
  exit 0
fi
if [ "$2" != "quick" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $O/gpu_tests.txt 2>&1
  tail -18 $O/gpu_tests.txt
fi
QIP_BENCH_DETAIL=$O/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"; tail -2 $O/bench_n1.err
QIP_BENCH_DETAIL=$O/bench_detail_2ranks.json QIP_BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --n-local 24 --steps 3 --warmup 1 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
tail -2 $O/bench_2ranks_one_gpu.err
timeout 600 python tools/bench_ops.py 30 all > $O/ops_table.md 2> $O/ops_table.err
timeout 600 python tools/bench_ops.py 30 all f32 > $O/ops_table_f32.md 2> $O/ops_table_f32.err
timeout 300 python tools/bench_permute.py 30 > $O/permute_f64.md 2>&1
timeout 300 python tools/bench_permute.py 30 f32 > $O/permute_f32.md 2>&1
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
if [ "$2" = "quick" ]; then  # the tests that cover what changed since the last full run of the suite (kept beside it under profiles/)
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 -k "sparse or wide_tiles or sharded_virtual or sharded_state or bench_multi_rank or every_timed_leg or (full_size_oracle_windows and 28)" > $O/gpu_tests_subset.txt 2>&1
  tail -14 $O/gpu_tests_subset.txt
fi
python - <<PY
import json
d=json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","gates_per_s","parity_ok","wall_s")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["parity"]["legs_failed"], d["parity"]["seconds"])
print(d["cpu_baseline"])
x=json.load(open("$O/bench_detail.json"))
for k,v in x.get("extras",{}).items():
    if isinstance(v,dict) and "ms" in v: print(k, round(v["ms"],1), v.get("launches"), round(v.get("per_launch_GBps",0)))
    elif isinstance(v,dict):
        print(k, {a:(round(b["ms"],1) if isinstance(b,dict) and "ms" in b else None) for a,b in v.items() if isinstance(b,dict)})
d=json.loads(open("$O/bench_2ranks_one_gpu.json").read().strip().splitlines()[-1])
print("2 ranks:", {k:d.get(k) for k in ("value","parity_ok","rccl_ranks","per_gpu_efficiency")})
PY
