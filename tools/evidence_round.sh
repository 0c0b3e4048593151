# Everything the round's profiles/ files come from, in one GPU call (run through gpurun):
#   bash tools/evidence_round.sh r03 [quick]
# results under gpurun_out/<tag>/ ; tools/summarize_rocprof.py and the copy into profiles/ run afterwards on the build box.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.txt 2>&1
  tail -15 $O/gpu_tests.txt
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
tail -2 $O/bench_n1.err
QIP_BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --n-local 24 --steps 3 --warmup 1 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
tail -2 $O/bench_2ranks_one_gpu.err
timeout 600 python tools/bench_ops.py 30 all > $O/ops_table.md 2> $O/ops_table.err
timeout 600 python tools/bench_ops.py 30 all f32 > $O/ops_table_f32.md 2> $O/ops_table_f32.err
QIP_SINGLE_VIA_TILE=0 QIP_SINGLE_VIA_TILE_F32=0 timeout 600 python tools/bench_ops.py 30 all > $O/ops_table_dedicated_kernels.md 2> $O/ops_table_dedicated_kernels.err
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
# PMC passes over the run-time-compiled tile sweeps of QFT alone (issue-bound, IEEE-equal) and of QFT with tile = 2 + fma + merged runs
rm -rf $R/gpurun_out/pmc_tileA $R/gpurun_out/pmc_tileB
QIP_TILE_JIT=1 bash tools/pmc_tile.sh qft 1 30 > $O/pmc_tile_qft.log 2>&1
python tools/pmc_tile_summary.py > $O/pmc_tile_qft_tile1.txt 2>&1
rm -rf $R/gpurun_out/pmc_tileA $R/gpurun_out/pmc_tileB
QIP_TILE_JIT=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 bash tools/pmc_tile.sh qft 2 30 >> $O/pmc_tile_qft.log 2>&1
python tools/pmc_tile_summary.py > $O/pmc_tile_qft_tile2_merge.txt 2>&1
rm -rf $R/gpurun_out/pmc_tileA $R/gpurun_out/pmc_tileB
cat $O/pmc_tile_qft_tile1.txt $O/pmc_tile_qft_tile2_merge.txt
python - <<PY
import json
d=json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","gates_per_s")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["parity"]["all_legs_ok"], d["parity"]["seconds"])
print(d["cpu_baseline"])
for k,v in d["extras"].items():
    if isinstance(v,dict) and "ms" in v: print(k, round(v["ms"],1), v.get("launches"), round(v.get("per_launch_GBps",0)))
    elif isinstance(v,dict):
        print(k, {a:(round(b["ms"],1) if isinstance(b,dict) and "ms" in b else None) for a,b in v.items() if isinstance(b,dict)})
PY
