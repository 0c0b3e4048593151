# the round's last measurement: the new scheduler defaults (tile_sched = 1: position search + pass-minimising gate order) on the benchmark
# circuits, then the run-time-compiled segments' bit-identity tests on the new plans (first-come numbers: tools/exp_r03y.sh, sched0.jsonl)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03z
mkdir -p $O
export QIP_TILE_JIT=1 QIP_TILE_RELABEL=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1
timeout 45 python tools/bench_tile.py 30 5 c2,c4,c2x4,grover,qft 1,2 > $O/sched1.jsonl 2> $O/sched1.err
python - <<'PY'
import json
for l in open("gpurun_out/r03z/sched1.jsonl"):
    d = json.loads(l)
    print(d["circuit"], d["tile"], d["sweeps"], d["ms"], d["norm"])
PY
unset QIP_TILE_JIT QIP_TILE_RELABEL QIP_TILE_FMA QIP_TILE_MERGE
timeout 45 python -m pytest tests/test_parity_gpu.py -q -x -k "compiled_at_run_time_are_bit_identical or merged_diagonal or builder_run_loop_uses_tile" 2>&1 | tail -4 | tee $O/tests.txt
