"""SURVEY.md §8 row e: the sharded state (virtual shards / several ranks on one GPU, RCCL plumbing at world = 1).
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


def test_sharded_virtual_shards_on_one_gpu():
    """world_size 2 on ONE GPU (world 4 is covered on CPU by tests/test_distributed_cpu.py)."""
    out = _run_dist(2, [])
    assert out.count("ok n=") == 4
    assert "ok fault: a failed exchange poisons the handle" in out and "ok pieces:" in out
    assert "ok fold: the remap's gather rides in the preceding tile sweep" in out
    assert "ok overlap: the exchange in slices beside the neighbouring tile sweeps changes nothing" in out
    assert "ok pair_floor on shards" in out
    assert out.count("samples differ from the reference's scan") == 2


@pytest.mark.slow
def test_sharded_state_against_the_oracle_at_bench_shard_size():
    """2 ranks x 2^27 amplitudes on ONE GPU (n = 28; r6: was 2^28 — the suite's time limit): the sharded path — localized ops, tile sweeps on the shards, k_pack_bits,
    the k_permute_bits route of a pack that gathers index bit 0, 2-D grids — checked against the oracle on closed sub-cubes of
    the LOGICAL index space read through the layout, with a twin sharded state on the literal kernel compared over all 2^28
    amplitudes after every step and closed-form marginals through qip_hip_dist_measure_probs (tests/dist_worker_parity_gpu.py)."""
    import json

    out = _run_dist(2, ["--n-local", "27", "--quick"], worker="dist_worker_parity_gpu.py", timeout=1800)
    res = json.loads([l for l in out.splitlines() if l.startswith("SHARDED_PARITY ")][-1][len("SHARDED_PARITY "):])
    assert res["all_legs_ok"] and res["n"] == 28 and res["rows_checked"] >= 10**7, res
    assert res["whole_vector"]["amplitudes_not_equal_in_IEEE_legs"] == 0 and res["bit_equal"], res


def test_sharded_rccl_plumbing_world1():
    out = _run_dist(1, ["--nccl"])
    assert out.count("ok n=") == 4


@pytest.mark.slow
def test_bench_multi_rank_code_path_on_one_gpu():
    """`python bench.py --gpus 2` as a PLAIN command (it re-launches itself as two ranks under torch.distributed.run):
    sharded state, plan/run_plan, remap, max-over-ranks timing, JSON line with the BASELINE configs[3]/[4] legs — with
    two ranks sharing the one GPU through the gloo / host-staged test hook."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    detail_path = os.path.join(tempfile.mkdtemp(), "bench_detail.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--n-local", "20", "--gates", "64", "--dist-overlap", "4", "--budget-s", "800"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, QIP_BENCH_DIST_BACKEND="gloo", QIP_BENCH_DETAIL=detail_path))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    # r6: the compact contract line comes three times — after the headline (parity pending), after the sharded parity check, at the end
    assert len(lines) == 3 and all(len(json.dumps(l)) < 4096 for l in lines), [len(json.dumps(l)) for l in lines]
    assert lines[0]["parity_ok"] is None and lines[0]["value"] > 0 and lines[1]["parity_ok"] is True and lines[0]["value"] == lines[2]["value"]
    line = lines[-1]
    assert res.stdout.rstrip().splitlines()[-1].startswith("{")  # ... and it is the LAST line of stdout
    assert line["n_gpus"] == 2 and line["config"]["n_qubits"] == 21 and line["scaling"] == "weak" and line["stage"] == "final"
    assert abs(line["norm_sqr_after"] - 1) < 1e-10
    # (the look-ahead keeps the qubit whose next H is farthest on the rank bit, and X gates there only rename the ranks:
    # a short headline circuit may need no exchange at all — the legs below and the parity leg do)
    assert line["value"] > 0 and line["roofline"]["kernel"].startswith("k_") and line["comm"]["remaps"] >= 0
    assert line["rccl_ranks"] == 0  # (the host-staged test transport: RCCL saw nobody)
    detail = json.load(open(detail_path))
    ex = detail["extras"]
    assert not detail.get("extras_skipped"), detail.get("extras_skipped")
    for name in ("configs3_clifford_t_n21", "configs4_grover_iteration_n21", "configs4_grover_dense_k3_n21",
                 "configs1_mixed_n21", "headline_tiled_mode1", "configs3_clifford_t_tiled_mode1", "configs1_mixed_tiled_mode1",
                 "configs1_mixed_tiled_mode1_jit_wide", "configs1_mixed_tiled_mode1_jit_wide_overlap", "configs1_mixed_tiled_mode1_overlap"):
        assert "error" not in ex[name] and ex[name]["ops_per_s"] > 0, (name, ex[name])
    assert sum(ex[name]["comm_over_reps"]["remaps"] for name in ("configs3_clifford_t_n21", "configs4_grover_iteration_n21", "configs1_mixed_n21")) >= 1
    assert abs(ex["norm_sqr_end"] - 1) < 1e-9
    par = detail["parity"]  # the sharded path against the oracle, inside the bench run itself
    assert "error" not in par and par["world"] == 2 and par["remaps_exercised"] >= 1 and par["max_abs_delta"] <= 1e-12, par
    # r4: the sharded state is checked at the size it was timed at (sub-cubes through the layout + twin + marginals), and the
    # verdict is a top-level field the run's exit status follows
    assert line["parity_ok"] is True and par["all_legs_ok"] and par["n"] == 21 and par["small_full_vector"]["ok"], par
    assert par["whole_vector"]["amplitudes_not_equal_in_IEEE_legs"] == 0 and par["packs_via_permute_bits"] >= 1, par
