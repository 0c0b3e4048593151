"""bench.py's contract line without a GPU (BENCH_r05.json had `parsed: null` — the one JSON line had grown to
24 KB).  The line is built by a pure function from a result dict: here from a canned result with everything at its worst case
(N = 8, every optional field present, non-finite numbers in the input)."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "parity_ok", "parity")


def canned(world=1, parity_ok=True):
    res = {
        "n_gpus": world, "steps": 20, "warmup": 5, "n_local": 30, "gates": 256, "n": 30 + int(math.log2(world)),
        "value": 6531.123456789 * world, "ms_per_step": 1346.7891234, "bytes_per_step": 256 * 32 * 2 ** (30 + int(math.log2(world))),
        "gates_per_s": 190.1234567, "norm_sqr": 0.9999999999999,
        "roofline": {"bound": "hbm", "kernel": "k_tile_passes", "achieved": 6513.77, "peak": 8000.0, "unit": "GB/s", "frac": 0.81422,
                     "traffic": 3.4381e10, "traffic_source": "static: profiles/r06_pmc_traffic.md (separate rocprofv3 --pmc passes of this command)",
                     "traffic_stale": False, "avg_launch_ms": 5.275, "launches": 2620, "algorithmic_bytes_per_launch": 34359738368.0},
        "cpu_baseline": {"value": 37.1, "unit": "GB/s", "cores": 16, "kind": "port", "gates_per_s": 4.3, "ms_per_gate": 231.5,
                         "sample": "first 8 gates of the headline circuit at n=28 (2 x 4 GiB), C restatement of qip-iterators 1.5.0 "
                                   "apply_op_overwrite, gcc -O3 -fopenmp, 16 threads, median of 3 (5.6 s CPU work)",
                         "detail": {"big": list(range(1000))}},
        "parity_ok": parity_ok,
        "parity": {"checker": "CPU oracle on closed sub-cubes + twin state over all 2^n amplitudes", "n": 30, "legs": 24, "legs_failed": [] if parity_ok else ["x" * 60] * 6,
                   "gates_checked": 2258, "rows_checked": 575275008, "max_abs_delta_IEEE_legs": 0.0, "max_abs_delta_1e-12_legs": float("nan"),
                   "whole_vector_compares": 141, "whole_vector_amplitudes_not_equal_IEEE_legs": 0, "seconds": 61.2},
        "extras_skipped": [{"section": "tolerance"}],
    }
    if world > 1:
        res.update({"rccl_ranks": world, "per_gpu_efficiency": 0.83123456, "per_gpu_efficiency_reference": {"value": 6531.0, "source": "profiles/n1_reference.json"},
                    "comm": {"remaps": 12, "pack_sweeps": 3, "bytes_sent": 1 << 40, "exchange_ms": float("inf"), "pack_ms": 61.0, "rccl_ranks": world, "rccl_rank": 0,
                             "pieces_sent": 84, "piece_bytes": 1 << 30, "packs_via_permute": 1, "packs_folded": 9, "remaps_overlapped": 0,
                             "remaps_overlapped_after": 0, "slices_overlapped": 0}})
    return res


def test_contract_line_is_compact_valid_json_with_the_required_keys():
    for world in (1, 2, 8):
        for ok in (True, False, None):
            text = bench.contract_line(dict(canned(world, ok), stage="final"))
            assert len(text) < bench.CONTRACT_MAX_BYTES and "\n" not in text, len(text)
            assert "NaN" not in text and "Infinity" not in text
            line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
            for key in REQUIRED:
                assert key in line, key
            assert line["n_gpus"] == world and line["unit"] == "GB/s" and line["dtype"] == "f64" and line["config"]["workload"]
            assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
            assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"]) and "detail" not in line["cpu_baseline"]
            if ok is False:  # a failed parity withholds the value
                assert line["value"] is None and line["value_withheld"] > 0 and line["parity_ok"] is False
            else:
                assert line["value"] > 0
            if world > 1:
                assert line["rccl_ranks"] == world and 0 < line["per_gpu_efficiency"] < 1 and line["comm"]["remaps"] == 12
                assert line["comm"]["exchange_ms"] is None  # (a non-finite number becomes null, never a bare token)


def test_bench_cli_contract_without_a_gpu():
    """the flags the driver passes parse; without a device the script fails loudly (no CPU fallback) instead of printing a line"""
    a = bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup, a.n_local, a.gates) == (1, 20, 5, 30, 256) and a.budget_s > 0
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, QIP_BENCH_DETAIL=os.devnull))
    assert p.returncode != 0 and "needs a HIP device" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
