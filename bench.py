#!/usr/bin/env python
"""bench.py — gate-application throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment re-launches itself as N ranks (one per GPU) under
torch.distributed.run on 127.0.0.1; under an existing launcher (RANK / WORLD_SIZE set) it runs as that rank.

A "step" is one pass of the hot path over one batch of synthetic input: 256 random single-qubit gates
(uniform over H / X / Rz(theta), uniform target — the single-qubit part of the configs[1] generator,
SURVEY.md §8(d)) applied to a 2^n-amplitude Complex<f64> state that is already resident in HBM.
n = n_local + log2(N) with n_local = 30 (the size BASELINE.json's target is quoted on: n = 30 on 1 GPU,
n = 33 on 8); weak scaling.  value = algorithmic GB/s of the whole job (32 * 2^n bytes per gate / wall time).

OUTPUT.  The LAST line of stdout is the compact CONTRACT LINE (one JSON object, < 4 KB): metric, value, unit, n_gpus,
steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline, parity_ok, a parity summary, and for N > 1 rccl_ranks /
per_gpu_efficiency / comm.  It is printed as soon as the headline, its parity check and the CPU baseline are done
(flushed), and printed again as the final line once the extras have run, so a problem in an extras leg cannot cost the
headline.  Everything else — every parity leg, per-kernel figures, the extras — goes to bench_detail.json beside this
file (rewritten after every leg).

Order of work (N = 1):
  1. parity, core   the timed configuration against the CPU ORACLE before anything is timed: seeded product state with
                    pairwise distinct amplitudes; the first 32 gates of the timed circuit and of the configs[1] mix gate
                    by gate — closed sub-cubes of 2^16+ rows vs the oracle's apply_op_overwrite / apply_op_row, a twin
                    state through the literal kernel compared over all 2^n amplitudes after every gate, closed-form
                    marginals (oracle/bench_parity.py, oracle/window_parity.py) — the checker, never timed
  2. headline       W warm-up steps, K timed steps between barriers; roofline of the dominant kernel from HIP events on
                    the launching stream inside the timed region
  3. cpu_baseline   the CPU oracle (C restatement of qip-iterators apply_op_overwrite, OpenMP over the host's cores) on a
                    bounded sample of the same circuit (rank 0, N = 1 only)
  4. contract line
  5. extras         the other BASELINE configs, the reference's own bench shapes, the optional modes; every mode is
                    checked against the oracle at the timed size right before it is timed; bounded by --budget-s of wall
                    time (legs that no longer fit are listed as skipped, never run unchecked).  A failed check is fatal
                    for the line: parity_ok false, value null, exit status 1.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
REPS = 5                # repetitions of every extras leg (median reported)
T_START = time.perf_counter()
DETAIL_PATH = os.environ.get("QIP_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
CONTRACT_MAX_BYTES = 4096


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-local", type=int, default=30, help="qubits per GPU shard (2^n_local amplitudes)")
    ap.add_argument("--gates", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="N = 1: only the timed headline steps (profiling passes: no parity check, extras or CPU baseline).  "
                         "N > 1: headline + rccl_ranks + the sharded parity check on two chunks, no extras (finishes in minutes)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--budget-s", type=float, default=270.0,
                    help="wall-clock budget of the whole run: extras legs that would not fit are skipped (and listed)")
    ap.add_argument("--deadline-s", type=float, default=0.0,
                    help="hard stop (0 = 3 x budget): a run still going then re-prints the contract line it has and exits, "
                         "or exits 2 without one — a hung collective must not hang the driver")
    ap.add_argument("--only", default="", help="comma-separated extras sections to run (default: all that fit)")
    ap.add_argument("--dist-overlap", type=int, default=0,
                    help="N > 1 only, experimental: add legs with the exchange cut into this many slices (option dist_overlap)")
    return ap.parse_args(argv)


def self_spawn(args) -> int:
    """`python bench.py --gpus N` as a plain command: become N ranks under torch.distributed.run."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------------
# the contract line (pure functions: tests/test_bench_contract.py builds one from a canned result without a GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _clean(x):
    """JSON-safe copy: NaN / +-Infinity become null (json.dumps would print bare NaN, which is not JSON)"""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if hasattr(x, "item") and not isinstance(x, (str, bytes)):
        try:
            return _clean(x.item())
        except Exception:
            return str(x)
    return x


def _sig(x, digits=6):
    return float(f"{x:.{digits}g}") if isinstance(x, float) and math.isfinite(x) else x


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_ok", "parity")


def contract_line(res: dict) -> str:
    """The compact line.  `res` holds: args (n_gpus, steps, warmup, n_local, gates), n, value, ms_per_step, bytes_per_step,
    gates_per_s, norm_sqr, roofline, cpu_baseline, parity_ok, parity (summary), stage, and for N > 1 comm / rccl_ranks /
    per_gpu_efficiency."""
    world = res["n_gpus"]
    g = int(math.log2(world))
    parity_ok = res.get("parity_ok")
    line = {
        "metric": "single-qubit gate apply GB/s (algorithmic bytes: 32 * 2^n per H / X / Rz gate)",
        "value": _sig(res["value"], 7) if parity_ok is not False else None, "unit": "GB/s",
        "n_gpus": world, "steps": res["steps"], "warmup": res["warmup"], "ms_per_step": _sig(res["ms_per_step"], 7),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{res['gates']} random single-qubit gates (uniform H / X / Rz(theta), uniform target; configs[1] generator, seed 28, "
                        f"single-qubit part) at n={res['n']} ({res['n_local']} qubits = {16 * 2**res['n_local'] / 2**30:.0f} GiB per GPU), "
                        f"Complex<f64>, resident non-uniform state",
            "n_qubits": res["n"], "n_local": res["n_local"], "gates_per_step": res["gates"],
            "algorithmic_bytes_per_step": res["bytes_per_step"],
            "parallelism": "single GPU" if world == 1 else f"state sharded by top {g} index bits over {world} GPUs, RCCL all-to-all qubit remap",
        },
        "gates_per_s": _sig(res.get("gates_per_s")), "frac_of_hbm_peak_per_gpu": _sig(res["value"] / world / HBM_PEAK_GBPS, 4),
        "norm_sqr_after": res.get("norm_sqr"),
        "roofline": None, "cpu_baseline": None,
        "parity_ok": parity_ok, "parity": res.get("parity"),
        "stage": res.get("stage", "final"), "detail_file": os.path.basename(DETAIL_PATH), "wall_s": round(time.perf_counter() - T_START, 1),
    }
    rf = res.get("roofline")
    if rf:
        line["roofline"] = {k: (_sig(v) if isinstance(v, float) else v) for k, v in rf.items()
                            if k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms", "launches",
                                     "algorithmic_bytes_per_launch", "traffic_source")}
    cpu = res.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {k: (_sig(v) if isinstance(v, float) else v) for k, v in cpu.items() if k != "detail"}
    if world > 1:
        line["rccl_ranks"] = res.get("rccl_ranks")
        line["per_gpu_efficiency"] = _sig(res.get("per_gpu_efficiency"), 4)
        line["per_gpu_efficiency_reference"] = res.get("per_gpu_efficiency_reference")
        line["comm"] = res.get("comm")
    if parity_ok is False:
        line["value_withheld"] = _sig(res["value"], 7)
    if res.get("extras_skipped"):
        line["extras_skipped"] = len(res["extras_skipped"])
    if res.get("truncated"):
        line["truncated"] = res["truncated"]
    text = json.dumps(_clean(line), allow_nan=False, separators=(",", ":"))
    if len(text) > CONTRACT_MAX_BYTES:  # cannot happen with the fields above; never let an oversized line out
        for k in ("comm", "per_gpu_efficiency_reference", "parity"):
            if isinstance(line.get(k), dict):
                line[k] = {"see": os.path.basename(DETAIL_PATH)}
        text = json.dumps(_clean(line), allow_nan=False, separators=(",", ":"))
    return text


class Emitter:
    """prints the contract line (flushed), remembers the last one for the watchdog, keeps bench_detail.json current"""

    def __init__(self, rank):
        self.rank = rank
        self.last = None
        self.res = None
        self.detail = {}
        self.lock = threading.Lock()

    def emit(self, res, stage):
        if self.rank != 0:
            return
        res = dict(res, stage=stage)
        text = contract_line(res)
        with self.lock:
            self.last, self.res = text, res
            print(text, flush=True)
        self.write_detail()

    def write_detail(self):
        if self.rank != 0:
            return
        try:
            body = dict(self.detail)
            if self.last:
                body = dict(json.loads(self.last), **body)
            tmp = DETAIL_PATH + ".tmp"
            with open(tmp, "w") as f:
                json.dump(_clean(body), f, indent=1)
            os.replace(tmp, DETAIL_PATH)
        except Exception as exc:  # noqa: BLE001 — the side file must never take the line down
            print(f"bench.py: could not write {DETAIL_PATH}: {exc!r}", file=sys.stderr)


def start_watchdog(em: Emitter, deadline_s: float):
    """A run that is still going at the deadline (a hung collective, a leg far over its estimate) ends NOW: the contract line it
    already has is printed once more, marked truncated, and the process exits 0; without one it exits 2 with a message."""
    def fire():
        with em.lock:
            if em.last and em.res is not None:
                try:
                    print(contract_line(dict(em.res, truncated=f"deadline of {deadline_s:.0f} s reached during '{em.detail.get('_now', '?')}'")), flush=True)
                except Exception:  # noqa: BLE001
                    print(em.last, flush=True)
                os._exit(0 if em.res.get("parity_ok") is not False else 1)
            print(f"bench.py: deadline of {deadline_s:.0f} s reached before the headline was measured (stage: {em.detail.get('_now', 'start-up')})",
                  file=sys.stderr, flush=True)
            os._exit(2)

    t = threading.Timer(deadline_s, fire)
    t.daemon = True
    t.start()
    return t


class Budget:
    def __init__(self, total_s):
        self.total = total_s

    def left(self):
        return self.total - (time.perf_counter() - T_START)


# ---------------------------------------------------------------------------------------------------------------------
def circuit_bytes(q, n, ops):
    return sum(q.algorithmic_bytes(n, op) for op in ops)


def dominant_kernel(profile):
    if not profile:
        return None, None
    name = max(profile, key=lambda k: profile[k]["total_ms"])
    return name, profile[name]


def load_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (profiles/): a STATIC
    figure from separate --pmc passes of this same command, not a per-run measurement.  Third value: True when the
    kernel source has changed since those passes were taken (the figure then describes older kernels)."""
    import hashlib

    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        now = hashlib.sha256(open(os.path.join(ROOT, "rustqip_amd", "csrc", "qip_kernels.h"), "rb").read()).hexdigest()[:16]
        stale = d.get("_kernels_sha16") != now
        return d.get(kernel_name, {}).get("hbm_bytes_per_launch"), d.get("_source", "profiles/pmc_traffic.json"), stale
    except Exception:
        return None, None, None


def load_n1_reference(n_local):
    """The single-GPU value of this same bench at the same shard size (profiles/n1_reference.json, written from a measured
    N = 1 line): the denominator of SURVEY.md §8(e)'s per-GPU efficiency."""
    try:
        with open(os.path.join(ROOT, "profiles", "n1_reference.json")) as f:
            return json.load(f).get(str(n_local))
    except Exception:
        return None


def median_time(fn, sync, reps=REPS):
    """fn() once untimed, then `reps` individually timed runs; returns (median seconds, all seconds)."""
    fn()
    sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts), ts


def st_device(st):
    return getattr(st, "device", 0)


def make_roofline(kname, kstat, n_local):
    per_launch_bytes = kstat["algorithmic_bytes"] / kstat["launches"]
    avg_ms = kstat["total_ms"] / kstat["launches"]
    achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_stale = load_traffic(kname)
    if n_local != 30 and traffic is not None:
        # the PMC passes were taken at the default size (2^30 amplitudes per GPU): a sweep's traffic is proportional to the shard
        traffic *= 2.0 ** (n_local - 30)
        traffic_src = f"{traffic_src}, scaled from 2^30 to 2^{n_local}"
    if traffic_stale:
        print("bench.py: warning: profiles/pmc_traffic.json was measured on an older csrc/qip_kernels.h — roofline.traffic is "
              "flagged stale; re-run tools/profile_round.sh", file=sys.stderr)
    return {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_source": f"static: {traffic_src} (separate rocprofv3 --pmc passes of this command)" if traffic_src else None,
            "traffic_stale": traffic_stale, "avg_launch_ms": avg_ms, "launches": kstat["launches"],
            "algorithmic_bytes_per_launch": per_launch_bytes}


# ---------------------------------------------------------------------------------------------------------------------
# extras (N = 1): every section = (name, estimated seconds, function); run in this order while the budget lasts
# ---------------------------------------------------------------------------------------------------------------------
def reference_bench_shapes(q, circuits, cpu_shape):
    """The reference's OWN benches (the only numbers its code defines: `cargo +nightly bench`), at the reference's sizes:
    qip/benches/state_bench.rs:118-139 (n = 8, dense 8-qubit H^8), :141-155 (n = 24, H on qubit 0), :157-170 (n = 8, 7-control
    identity), :172-202 (n = 16, 15-control identity, Complex<f64> and Complex<f32>), :380-393 (n = 16, 16-qubit sparse identity);
    qip-iterators/benches/matmul_bench.rs:163-177 (n = 20, [1,1,1,1] on qubit 0; real f64 there, complex here).  One `apply_op`
    per iteration there, with the op built once outside the loop; here the op is PREPARED once (a program owns the device copy
    of its tables) and applied three ways on a resident state — eager (one C-ABI call = the op's kernel launches, nothing else),
    a hipGraph program of 64 ops, the same program over tile sweeps (tile = 1) — beside the CPU restatement's apply_op
    (accumulate, like the reference's bench loop) on the host's cores.  These states live in L2 / MALL: launch-bound."""
    import numpy as np

    isq = 1.0 / math.sqrt(2.0)
    h = [isq, isq, isq, -isq]
    h8 = np.array([[1.0]])
    for _ in range(8):
        h8 = np.kron(h8, np.array([[isq, isq], [isq, -isq]]))
    shapes = [
        ("state_bench.rs:118-139 bench_hadamard_larger", 8, np.complex128, q.make_matrix_op(list(range(8)), h8.ravel())),
        ("state_bench.rs:141-155 bench_hadamard_larger_single", 24, np.complex128, q.make_matrix_op([0], h)),
        ("state_bench.rs:157-170 bench_cidentity_larger", 8, np.complex128, q.make_control_op(list(range(7)), q.make_matrix_op([7], [1, 0, 0, 1]))),
        ("state_bench.rs:172-186 bench_cidentity_giant", 16, np.complex128, q.make_control_op(list(range(15)), q.make_matrix_op([15], [1, 0, 0, 1]))),
        ("state_bench.rs:188-202 bench_cidentity_giant_halfprec", 16, np.complex64, q.make_control_op(list(range(15)), q.make_matrix_op([15], [1, 0, 0, 1]))),
        ("state_bench.rs:380-393 bench_identity_giant_sparse", 16, np.complex128, q.make_sparse_matrix_op(list(range(16)), [[(i, 1.0)] for i in range(1 << 16)])),
        ("matmul_bench.rs:163-177 bench_large_ones_qip", 20, np.complex128, q.make_matrix_op([0], [1, 1, 1, 1])),
    ]
    out = {}
    reps = 64
    for name, n, dtype, op in shapes:
        row = {"n": n, "dtype": "c64" if dtype == np.complex128 else "c32", "algorithmic_bytes_per_op": q.algorithmic_bytes(n, op, 0 if dtype == np.complex128 else 1)}
        try:
            with q.HipState(n, dtype) as s:
                s.init_basis(0)
                one = s.compile_program([op])
                one.run()
                s.sync()
                ts = []
                for _ in range(REPS):
                    s.sync()
                    t = time.perf_counter()
                    for _ in range(reps):
                        one.run()
                    s.sync()
                    ts.append((time.perf_counter() - t) / reps)
                row["eager_us_per_op"] = 1e6 * statistics.median(ts)
                one.close()
                # (what a caller pays who builds the op anew for every application, like LocalBuilder's run loop does: pack + upload + launch)
                s.apply_ops([op])
                s.sync()
                t = time.perf_counter()
                for _ in range(8):
                    s.apply_ops([op])
                s.sync()
                row["one_shot_us_per_op"] = 1e6 * (time.perf_counter() - t) / 8
                for label, tile in (("hipgraph_program_us_per_op", 0), ("tiled_program_us_per_op", 1)):
                    if tile and n < 11:
                        continue  # (a tile is 2^11 amplitudes)
                    s.set_option("tile", tile)
                    prog = s.compile_program([op] * reps)
                    prog.run()
                    s.sync()
                    ts = []
                    for _ in range(REPS):
                        s.sync()
                        t = time.perf_counter()
                        prog.run()
                        s.sync()
                        ts.append((time.perf_counter() - t) / reps)
                    row[label] = 1e6 * statistics.median(ts)
                    row[label.replace("_us_per_op", "_is_graph")] = bool(prog.is_graph)
                    prog.close()
                    s.set_option("tile", 0)
        except Exception as exc:  # noqa: BLE001
            row["error"] = repr(exc)
        if cpu_shape is not None:
            try:
                row.update(cpu_shape(n, dtype, op))
            except Exception as exc:  # noqa: BLE001
                row["cpu_error"] = repr(exc)
        out[name] = row
    return out


def run_extras(q, circuits, st, n, args, ops, ops_mixed, par, em, budget, res):
    """extras on one GPU; `par` (oracle/bench_parity.Parity or None) checks a mode right before it is timed"""
    import ctypes as _C

    import numpy as np

    from rustqip_amd import _ffi as _F

    extras = em.detail.setdefault("extras", {})
    skipped = em.detail.setdefault("extras_skipped", [])
    seconds = em.detail.setdefault("extras_seconds", {})
    gates = args.gates

    def leg(cops, label_ops="gates", state=None, n_=None, **options):
        """median of REPS timed applications of a circuit with the given state options"""
        s_ = st if state is None else state
        for k, v in options.items():
            s_.set_option(k, v)
        cc = s_.compile_ops(cops)
        s_.set_option("profile", 1)
        s_.apply_compiled(cc)
        s_.sync()
        s_.profile_reset()
        dt, ts = median_time(lambda: s_.apply_compiled(cc), s_.sync)
        prof = s_.profile()
        s_.set_option("profile", 0)
        for k in options:
            s_.set_option(k, {"tile_auto": 1, "pair_floor": 1, "mfma": 1}.get(k, 0))
        sweeps = sum(v["launches"] for v in prof.values()) // (REPS + 1)
        sweep_bytes = sum(v["algorithmic_bytes"] for v in prof.values()) / (REPS + 1)
        by = circuit_bytes(q, n_ or n, cops)
        return {label_ops: len(cops), "ms": 1e3 * dt, "ms_min_max": [round(1e3 * min(ts), 3), round(1e3 * max(ts), 3)],
                "%s_per_s" % label_ops: len(cops) / dt, "algorithmic_GBps": by / dt / 1e9, "launches": sweeps,
                "per_launch_GBps": sweep_bytes / dt / 1e9, "reps": REPS, "options": options}

    def checked(name, cops, exact, seed, max_len=64, **options):
        """the mode's parity leg at the timed size; False (and the timed leg is not run) when it fails.  --no-parity (profiling
        runs): nothing is checked and the legs are timed as they are."""
        if par is None:
            return True
        return par.leg(name, cops, exact, seed=seed, max_len=max_len, **options)

    def jit_stats():
        c = _F.jit_counters()  # (cache misses of this process: compiled here, by a helper, or found on disk)
        return int(c["kernels_resident_total"]), c["compile_ms"] + c["disk_load_ms"]

    def timed_mode(dst, key, cops, label_ops, exact, seed, chunk, max_len=64, **options):
        """check `chunk` (a slice of the circuit at the timed size) through the mode, then time the whole circuit in it"""
        if not checked("%s__%s" % (dst if isinstance(dst, str) else "mixed", key), chunk, exact, seed, max_len=max_len, **options):
            return None
        k0, ms0 = jit_stats()
        r = leg(cops, label_ops, **options)
        k1, ms1 = jit_stats()
        if options.get("tile_jit"):
            r.update({"segments_compiled": k1 - k0, "compile_ms_once": ms1 - ms0})
        r["bar"] = "IEEE-equal" if exact else "1e-12"
        return r

    qft = circuits.c3_qft(n)
    cliff = circuits.c4_clifford_t(n, gates, seed=32)
    grover = circuits.c5_grover_iteration(n)
    grover3 = circuits.c5_grover_iteration(n, dense_k3=True)
    more = circuits.c2_random_circuit(n, 192, seed=29)

    def sec_mixed():
        # configs[1]'s mix, one launch per gate (checked in the core block), and the per-target H sweep
        mixed = leg(ops_mixed)
        mixed["frac_of_8TBps"] = mixed["algorithmic_GBps"] / HBM_PEAK_GBPS
        mixed["workload"] = "configs[1] generator: 3/4 H/X/Rz + 1/4 CNOT, seed 28, %d gates, n=%d" % (len(ops_mixed), n)
        em.detail["mixed_circuit"] = mixed
        sweep = []
        for tq in range(n):
            op = st.compile_ops([q.make_matrix_op([tq], circuits.H)] * 2)
            dt, _ = median_time(lambda: st.apply_compiled(op), st.sync, reps=3)
            sweep.append(round(32.0 * 2**n / (dt / 2) / 1e9, 1))
        extras["h_sweep_GBps_by_target_qubit"] = sweep
        extras["h_sweep_min_median_GBps"] = [min(sweep), float(np.median(sweep))]

    def sec_shapes():
        cpu_shape = None
        if not args.no_cpu_baseline:
            from oracle.bench_parity import cpu_shape
        extras["reference_bench_shapes"] = reference_bench_shapes(q, circuits, cpu_shape)

    def sec_generic_p():
        # the kernel's generic element type (matrix_ops.rs:98-107): apply_op<P> for a REAL P on device slices — the reference's own
        # f64 bench shapes (qip-iterators/benches/matmul_bench.rs:19-33 n = 12, :163-177 n = 20: a 2 x 2 matrix of ones on qubit 0,
        # ones in, accumulate) and the same op at an HBM size; every row bit-equal to the oracle's real restatement or the leg fails
        import torch

        rows = []
        ones = q.MatrixOp.new_matrix([0], [1.0, 1.0, 1.0, 1.0])
        for n_, dt, reps in ((12, np.float64, 200), (20, np.float64, 200), (26, np.float64, 20), (26, np.float32, 20)):
            x = np.ones(1 << n_, dtype=dt)
            d_in = torch.from_numpy(x).cuda(st_device(st))
            d_out = torch.zeros(1 << n_, dtype=d_in.dtype, device=d_in.device)
            row = {"n": n_, "P": np.dtype(dt).name, "op": "Matrix([0], ones), accumulate", "ref": "matmul_bench.rs:%s" % ("19-33" if n_ == 12 else "163-177" if n_ == 20 else "- (HBM size)")}
            if par is not None:
                from oracle import qip_oracle as _O  # (the checker)
                want = np.zeros(1 << n_, dtype=dt)
                for _ in range(2):
                    _O.apply_op(n_, ones, x, want)
                    q.apply_op_device(n_, ones, d_in, d_out)
                torch.cuda.synchronize()
                row["bit_equal_to_oracle"] = par.record("generic_p__%s_n%d" % (row["P"], n_), d_out.cpu().numpy(), want, gates=2)
            cop = ones.to_c(_F.QIP_F64 if dt == np.float64 else _F.QIP_F32)  # (the reference builds its op once, outside b.iter)
            q.apply_op_device(n_, cop, d_in, d_out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # (the call launches on the null stream = torch's current stream here)
            for _ in range(reps):
                q.apply_op_device(n_, cop, d_in, d_out)
            e1.record()
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / reps
            by = np.dtype(dt).itemsize * 3 * (1 << n_)  # input read + output read + output write
            row.update({"us_per_call": 1e6 * sec, "algorithmic_GBps": by / sec / 1e9, "frac_of_8TBps": by / sec / 1e9 / HBM_PEAK_GBPS})
            rows.append(row)
            del d_in, d_out
        extras["generic_p_real_vectors"] = rows

    def sec_builder():
        # What a `calculate_state` caller gets — HipBuilder's run loop: ONE apply_ops batch on a fresh handle with tile = 1 and
        # tile_relabel = 1, no program object (qip/src/builder.rs:499 is the call being replaced).  Option tile_auto: the interpreter
        # on a cold cache (the plan's segments go to background helpers), compiled wide sweeps once every segment is a memory /
        # disk-cache hit.  Checked in this process (interpreter + relabelling), then timed here cold, and — once the helpers have
        # delivered — in a SECOND PROCESS (tools/builder_one_shot.py), which is what "the next run of the user's program" means.
        import subprocess

        if not checked("builder__tile1_relabel", ops_mixed[32:96], True, 13, tile=1, tile_relabel=1, tile_auto=0):
            return
        r = {"options": {"tile": 1, "tile_relabel": 1}, "first_process_jit_before": _F.jit_counters()}
        cc = st.compile_ops(ops_mixed)
        st.set_option("tile", 1)
        st.set_option("tile_relabel", 1)
        try:
            st.sync()
            t = time.perf_counter()
            st.apply_compiled(cc)
            st.sync()
            r["first_call_ms"] = 1e3 * (time.perf_counter() - t)
            c = _F.jit_counters()
            r["first_call_took"] = "compiled sweeps (every segment was already cached)" if c["kernels_resident_total"] > r["first_process_jit_before"]["kernels_resident_total"] \
                else "interpreter (%d segments handed to background helpers)" % (c["background_segments"] - r["first_process_jit_before"]["background_segments"])
            dt, ts = median_time(lambda: st.apply_compiled(cc), st.sync, reps=3)
            r["interpreter_or_cached_ms_median3"] = 1e3 * dt
        finally:
            st.set_option("tile", 0)
            st.set_option("tile_relabel", 0)
        # the helpers' code objects: wait for them (bounded), then a second process
        want = _F.jit_counters()["background_segments"] - r["first_process_jit_before"]["background_segments"]
        cache_dir = _F.lib.qip_hip_jit_cache_dir().decode()
        t_wait = time.perf_counter()
        while cache_dir and want and time.perf_counter() - t_wait < 40.0:
            if not [f for f in os.listdir(cache_dir) if f.startswith("seg.")]:  # (the helpers remove their sources as they finish)
                break
            time.sleep(0.25)
        r["waited_for_background_s"] = round(time.perf_counter() - t_wait, 1)
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "builder_one_shot.py"), str(n)], capture_output=True, text=True, timeout=120)
            r["second_process"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": p.stderr[-400:]}
        except Exception as exc:  # noqa: BLE001
            r["second_process"] = {"error": repr(exc)}
        extras["builder_one_shot"] = r

    def sec_tiled():
        d = {}
        d["tile1_jit"] = timed_mode("mixed", "tile1_jit", ops_mixed, "gates", True, 14, ops_mixed[96:160], tile=1, tile_jit=1)
        d["tile1_jit_wide"] = timed_mode("mixed", "tile1_jit_wide", ops_mixed, "gates", True, 26, more[:64], tile=1, tile_jit=1, tile_wide=1)
        swaps = [q.make_swap_op([3], [n - 2]), q.make_swap_op([n - 9], [0])]  # (two Swap ops ride along as label exchanges in the checked chunk)
        if checked("mixed__tile1_jit_wide_relabel2", more[64:96] + swaps + more[96:128], True, 27, tile=1, tile_jit=1, tile_wide=1, tile_relabel=2):
            d["tile1_jit_wide_relabel"] = timed_mode("mixed", "tile1_jit_wide_relabel", ops_mixed, "gates", True, 15, ops_mixed[160:224],
                                                     tile=1, tile_jit=1, tile_wide=1, tile_relabel=1)
        extras["tiled"] = {k: v for k, v in d.items() if v}
        try:  # a PROGRAM created on a tile = 1 state (option tile_auto): compiled once at creation, replayed as one hipGraph
            st.set_option("tile", 1)
            t_c = time.perf_counter()
            prog = st.compile_program(ops_mixed)
            create_s = time.perf_counter() - t_c
            dt, ts = median_time(prog.run, st.sync)
            extras["program_tile_auto"] = {"gates": len(ops_mixed), "ms": 1e3 * dt, "gates_per_s": len(ops_mixed) / dt, "is_graph": bool(prog.is_graph),
                                           "create_s_once": create_s, "reps": REPS}
            prog.close()
        except Exception as exc:  # noqa: BLE001
            extras["program_tile_auto"] = {"error": repr(exc)}
        finally:
            st.set_option("tile", 0)

    def sec_config(cname, cops, chunks, wide, seed, max_len):
        """one BASELINE circuit: one launch per op, interpreter sweeps, compiled sweeps, compiled wide sweeps — each mode checked on
        its own slice of the circuit (chunks = [interpreter, compiled, wide, wide + relabel]) at the timed size first"""
        def run():
            d = leg(cops, "ops")  # one launch per op (the core block's path)
            d["tile1"] = timed_mode(cname, "tile1", cops, "ops", True, seed, chunks[0], max_len=max_len, tile=1)
            d["tile1_jit"] = timed_mode(cname, "tile1_jit", cops, "ops", True, seed + 1, chunks[1], max_len=max_len, tile=1, tile_jit=1)
            if wide:
                d["tile1_jit_wide"] = timed_mode(cname, "tile1_jit_wide", cops, "ops", wide == "exact", seed + 2, chunks[2], max_len=max_len,
                                                 tile=1, tile_jit=1, tile_wide=1)
            if len(chunks) > 3:
                d["tile1_jit_wide_relabel"] = timed_mode(cname, "tile1_jit_wide_relabel", cops, "ops", True, seed + 3, chunks[3], tile=1, tile_jit=1,
                                                         tile_wide=1, tile_relabel=1)
            extras[cname] = {k: v for k, v in d.items() if v is not None}
        return run

    def sec_n28():
        n28 = 28  # configs[1] exactly
        ops28 = circuits.c2_random_circuit(n28, gates, seed=28)
        with q.HipState(n28) as s28:
            s28.init_basis(0)
            s28.apply_ops(circuits.h_layer(n28) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n28)])
            c28 = s28.compile_ops(ops28)
            dt, ts = median_time(lambda: s28.apply_compiled(c28), s28.sync)
            extras["configs1_n28"] = {"GBps": circuit_bytes(q, n28, ops28) / dt / 1e9, "gates_per_s": len(ops28) / dt,
                                      "ms_per_step": 1e3 * dt, "reps": REPS, "norm_sqr": s28.norm_sqr()}

    def sec_f32():
        # the headline circuit on a Complex<f32> state (8 GiB at n = 30; 16 * 2^n bytes per gate), gate by
        # gate and as wide tile sweeps; the wide leg is checked against the f32 ORACLE on closed sub-cubes first (bit equality)
        with q.HipState(n, np.complex64, device=st_device(st)) as s32:
            s32.init_basis(0)
            s32.apply_ops(par.ops0 if par is not None else circuits.h_layer(n))
            okw = par.leg("complex64__tile1_jit_wide", ops_mixed[:64], True, seed=33, state=s32, tile=1, tile_jit=1, tile_wide=1) if par is not None else True
            c32 = s32.compile_ops(ops)
            dt, _ = median_time(lambda: s32.apply_compiled(c32), s32.sync)
            by32 = sum(q.algorithmic_bytes(n, op, 1) for op in ops)
            f32 = {"gates": len(ops), "ms": 1e3 * dt, "gates_per_s": len(ops) / dt, "algorithmic_GBps": by32 / dt / 1e9,
                   "frac_of_8TBps": by32 / dt / 1e9 / HBM_PEAK_GBPS, "reps": REPS}
            if okw:
                f32["mixed_tile1_jit_wide"] = leg(ops_mixed, state=s32, tile=1, tile_jit=1, tile_wide=1)
            f32["norm_sqr"] = s32.norm_sqr()
            extras["complex64_n%d" % n] = f32

    def sec_tol():
        # the 1e-12 modes: commuting reorder with fused multiply-adds and merged diagonal runs (compiled), dense fusion — 32-gate chunks
        d = {}
        d["tile2_jit_fma_merge_wide"] = timed_mode("mixed", "tile2_jit_fma_merge_wide", ops_mixed, "gates", False, 30, more[128:160],
                                                   tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1)
        d["fused_k5"] = timed_mode("mixed", "fuse5", ops_mixed, "gates", False, 19, more[160:192], fuse=5)
        d["qft_tile2_jit_fma_merge"] = timed_mode("qft", "tile2_jit_fma_merge", qft, "ops", False, 24, qft[:160], max_len=160,
                                                  tile=2, tile_jit=1, tile_fma=1, tile_merge=1)
        d["clifford_t_tile2_jit_fma_merge_wide_relabel"] = timed_mode("clifford", "tile2_jit_fma_merge_wide_relabel", cliff, "ops", False, 32, cliff[:32],
                                                                      tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1, tile_relabel=1)
        extras["tolerance_modes_1e-12"] = {k: v for k, v in d.items() if v}

    sections = [
        ("mixed", 6, sec_mixed),
        ("shapes", 12, sec_shapes),
        ("generic_p", 6, sec_generic_p),
        ("builder", 25, sec_builder),
        ("tiled", 30, sec_tiled),
        # (QFT's controlled phases only TEST their bits: a chunk is closed over its H targets alone and holds what a timed segment holds;
        #  the issue-bound QFT gains nothing from wide tiles; Clifford+T is the circuit relabelling pays for)
        ("configs2_qft_n%d" % n, 30, sec_config("configs2_qft_n%d" % n, qft, [qft[:240], qft[240:]], None, 40, 160)),
        ("configs3_clifford_t_n%d" % n, 25, sec_config("configs3_clifford_t_n%d" % n, cliff, [cliff[:64], cliff[64:128], cliff[128:192], cliff[192:]], "exact", 50, 64)),
        ("configs4_grover_iteration_n%d" % n, 25, sec_config("configs4_grover_iteration_n%d" % n, grover, [grover[:50], grover[50:100], grover[100:]], "exact", 60, 96)),
        ("configs4_grover_dense_k3_n%d" % n, 25, sec_config("configs4_grover_dense_k3_n%d" % n, grover3, [grover3[:40], grover3[40:80], grover3[80:]], "1e-12", 70, 96)),
        ("n28", 6, sec_n28),
        ("f32", 12, sec_f32),
        ("tolerance", 30, sec_tol),
    ]
    only = [x for x in args.only.split(",") if x]
    for name, est, fn in sections:
        if only and not any(name.startswith(o) for o in only):
            continue
        if budget.left() < est:
            skipped.append({"section": name, "estimated_s": est, "budget_left_s": round(budget.left(), 1)})
            continue
        em.detail["_now"] = "extras/" + name
        t0 = time.perf_counter()
        try:
            fn()
        except Exception as exc:  # noqa: BLE001 — an extras leg never takes the line down (a parity FAILURE is not an exception: see below)
            extras[name + "_error"] = repr(exc)
        seconds[name] = round(time.perf_counter() - t0, 1)
        if par is not None:
            em.detail["parity"] = par.detail()
            res["parity"] = par.summary()
            res["parity_ok"] = par.ok()
        em.write_detail()
        if par is not None and not par.ok():
            break  # fatal: main() prints the withheld line and exits 1
    res["extras_skipped"] = skipped
    extras["jit"] = dict(_F.jit_counters(), cache_dir=_F.lib.qip_hip_jit_cache_dir().decode())
    try:
        extras["norm_sqr_end"] = st.norm_sqr()
    except Exception as exc:  # noqa: BLE001
        extras["norm_sqr_end"] = repr(exc)


def run_dist_extras(q, circuits, st, n, args, ops, ops_mixed, em, budget, barrier, max_over_ranks, sync):
    """BASELINE configs[3] (Clifford+T) and configs[4] (Grover iteration, plain and dense k = 3) on the sharded state, and the
    headline / the mix with the local runs between remaps applied as tile sweeps.  Median of REPS, max over ranks; every leg
    is guarded, and skipped once the budget is used up (the decision is rank 0's, broadcast: the legs are collective)."""
    extras = em.detail.setdefault("extras", {})
    skipped = em.detail.setdefault("extras_skipped", [])

    def dist_leg(cops, tile=0, jit=0, wide=0, overlap=0):
        st.set_option("tile", tile)
        st.set_option("tile_jit", jit)
        st.set_option("tile_wide", wide)
        st.set_option("dist_overlap", overlap)
        cc = st.compile_ops(cops)
        st.apply_compiled(cc)
        sync()
        st.comm_stats()  # reset the counters
        ts = []
        for _ in range(REPS):
            barrier()
            t = time.perf_counter()
            st.apply_compiled(cc)
            sync()
            barrier()
            ts.append(max_over_ranks(time.perf_counter() - t))
        for key in ("tile", "tile_jit", "tile_wide", "dist_overlap"):
            st.set_option(key, 0)
        dt = statistics.median(ts)
        return {"ops": len(cops), "ms": 1e3 * dt, "ops_per_s": len(cops) / dt, "algorithmic_GBps": circuit_bytes(q, n, cops) / dt / 1e9,
                "reps": REPS, "comm_over_reps": st.comm_stats()}

    cliff = circuits.c4_clifford_t(n, args.gates, seed=32)
    legs = [("configs3_clifford_t_n%d" % n, cliff, {}),
            ("configs4_grover_iteration_n%d" % n, circuits.c5_grover_iteration(n), {}),
            ("configs4_grover_dense_k3_n%d" % n, circuits.c5_grover_iteration(n, dense_k3=True), {}),
            ("configs1_mixed_n%d" % n, ops_mixed, {}),
            ("configs3_clifford_t_tiled_mode1", cliff, {"tile": 1}),
            ("configs1_mixed_tiled_mode1", ops_mixed, {"tile": 1}),
            ("headline_tiled_mode1", ops, {"tile": 1}),
            ("configs1_mixed_tiled_mode1_jit_wide", ops_mixed, {"tile": 1, "jit": 1, "wide": 1}),
            ("configs3_clifford_t_tiled_mode1_jit_wide", cliff, {"tile": 1, "jit": 1, "wide": 1})]
    if args.dist_overlap >= 2:
        legs += [("configs1_mixed_tiled_mode1_jit_wide_overlap", ops_mixed, {"tile": 1, "jit": 1, "wide": 1, "overlap": args.dist_overlap}),
                 ("configs1_mixed_tiled_mode1_overlap", ops_mixed, {"tile": 1, "overlap": args.dist_overlap})]
    for cname, cops, kw in legs:
        go = max_over_ranks(0.0 if budget.left() >= (40 if kw.get("jit") else 15) else 1.0) == 0.0
        if not go:
            skipped.append({"section": cname, "budget_left_s": round(budget.left(), 1)})
            continue
        em.detail["_now"] = "extras/" + cname
        try:
            extras[cname] = dist_leg(cops, **kw)
        except Exception as exc:  # noqa: BLE001
            extras[cname] = {"error": repr(exc)}
        em.write_detail()
    try:
        extras["norm_sqr_end"] = st.norm_sqr()
    except Exception as exc:  # noqa: BLE001
        extras["norm_sqr_end"] = repr(exc)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    g = int(math.log2(world))
    if 1 << g != world:
        raise SystemExit("number of GPUs must be a power of two")
    if args.headline_only:
        args.no_extras = True
        if world == 1:
            args.no_parity = args.no_cpu_baseline = True

    em = Emitter(rank)
    budget = Budget(args.budget_s)
    start_watchdog(em, args.deadline_s if args.deadline_s > 0 else 3.0 * args.budget_s)
    em.detail["_now"] = "start-up"

    import numpy as np
    import torch

    import rustqip_amd as q
    from rustqip_amd import circuits

    if not torch.cuda.is_available() or q.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: rustqip_amd has no CPU fallback")
    # QIP_BENCH_DIST_BACKEND=gloo is a TEST hook: several ranks share one GPU and the remap all-to-all is
    # staged through host memory, so the N > 1 code path can be exercised where only one GPU exists.
    # Real multi-GPU runs use RCCL on device buffers.
    dist_backend = os.environ.get("QIP_BENCH_DIST_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import datetime

        import torch.distributed as dist

        kw = {"timeout": datetime.timedelta(seconds=max(120.0, args.budget_s))}  # (a rank that never arrives fails the group instead of hanging it)
        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device), **kw)
        else:
            dist.init_process_group(backend=dist_backend, **kw)

    n = args.n_local + g
    ops = circuits.c2_random_circuit(n, args.gates, seed=28, single_only=True)  # the headline: H / X / Rz only
    ops_mixed = circuits.c2_random_circuit(n, args.gates, seed=28)              # configs[1]: 3/4 of those + 1/4 CNOT
    bytes_per_step = circuit_bytes(q, n, ops)  # whole job (all ranks)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    par = None
    if world == 1:
        st = q.HipState(n, np.complex128, device=device)
        if not args.no_parity:
            from oracle.bench_parity import Parity

            em.detail["_now"] = "parity/core"
            par = Parity(q, st, n, device=device)
            par.core(ops, ops_mixed)
            par.reset_state()
            em.detail["parity"] = par.detail()
        else:  # same resident state as the checked run (seeded product state), without the oracle comparison
            from oracle import window_parity as W

            st.init_basis(0)
            st.apply_ops(W.product_state_ops(n, seed=n)[0])
        compiled = st.compile_ops(ops)
        run_step = lambda: st.apply_compiled(compiled)  # noqa: E731
        sync = st.sync
        set_profile = lambda v: st.set_option("profile", v)  # noqa: E731
        get_profile = lambda: (st.profile(), st.profile_reset())[0]  # noqa: E731
    else:
        from rustqip_amd.sharded import DistState

        # the sharded state inside libqip_hip.so (C ABI qip_hip_dist_*): planner, pack sweep and the RCCL exchange
        st = DistState(n, dist, device, np.complex128, host_staged=dist_backend != "nccl")
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
        st.comm_stats()  # reset: the headline's comm figures cover warm-up + timed steps only
        compiled = st.compile_ops(ops)
        run_step = lambda: st.apply_compiled(compiled)  # noqa: E731
        sync = st.sync
        set_profile = st.set_profile
        get_profile = st.take_profile

    em.detail["_now"] = "headline"
    for _ in range(args.warmup):
        run_step()
    sync()
    set_profile(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = max_over_ranks(t1 - t0)
    profile = get_profile()
    set_profile(0)
    comm_headline = st.comm_stats() if world > 1 else None
    rccl_ranks_min = None
    if world > 1:
        rr = torch.tensor([float(comm_headline["rccl_ranks"])], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(rr, op=dist.ReduceOp.MIN)
        rccl_ranks_min = int(rr.item())
    norm = st.norm_sqr()

    value = bytes_per_step * args.steps / elapsed / 1e9
    kname, kstat = dominant_kernel(profile)
    res = {
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "n_local": args.n_local, "gates": args.gates, "n": n,
        "value": value, "ms_per_step": 1e3 * elapsed / args.steps, "bytes_per_step": bytes_per_step,
        "gates_per_s": args.gates * args.steps / elapsed, "norm_sqr": norm,
        "roofline": make_roofline(kname, kstat, args.n_local) if kstat else None,
        "cpu_baseline": None,
        "parity_ok": par.ok() if par is not None else None,
        "parity": par.summary() if par is not None else ({"status": "pending"} if world > 1 and not args.no_parity else None),
    }
    em.detail["kernels"] = {
        k: {"launches": v["launches"], "avg_ms": v["total_ms"] / v["launches"],
            "GBps": (v["algorithmic_bytes"] / v["launches"]) / (v["total_ms"] / v["launches"] * 1e-3) / 1e9}
        for k, v in (profile or {}).items() if v["launches"] and v["total_ms"] > 0
    }
    if world > 1:
        res["comm"] = comm_headline
        res["rccl_ranks"] = rccl_ranks_min
        em.detail["dist"] = st.describe()
        ref = load_n1_reference(args.n_local)
        if ref:
            # (aggregate GB/s / G) / single-GPU GB/s at the same shard size, communication included
            res["per_gpu_efficiency"] = value / world / ref["value"]
            res["per_gpu_efficiency_reference"] = ref
        em.emit(res, "headline (sharded parity check pending)")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        em.detail["_now"] = "cpu_baseline"
        from oracle.bench_parity import cpu_baseline

        cpu = cpu_baseline(q, circuits, args.cpu_budget_s)
        em.detail["cpu_baseline_detail"] = cpu.pop("detail", None)
        res["cpu_baseline"] = cpu
    if world == 1:
        em.emit(res, "headline" if not args.no_extras else "final")

    # ---- after the line: extras (N = 1), sharded parity + extras (N > 1) -------------------------------------------------
    if world == 1 and not args.no_extras and res["parity_ok"] is not False:
        run_extras(q, circuits, st, n, args, ops, ops_mixed, par, em, budget, res)
    if world > 1 and not args.no_parity:
        # The N > 1 path against the CPU oracle on THIS fabric (the real transport, every rank's real kernels), at the size
        # that was just timed: closed sub-cubes of the logical index space gathered through the layout, a twin sharded state
        # on the literal kernel compared over all 2^n amplitudes after every step, closed-form marginals of the product state
        # (oracle/window_parity.sharded_parity) — plus a small sharded state compared as a gathered full vector.  The
        # checker, never timed.  A failure here is fatal for the line (parity_ok false, value null, rc 1).
        em.detail["_now"] = "parity/sharded"
        try:
            from oracle import qip_oracle as O
            from oracle import window_parity as W
            from rustqip_amd.sharded import DistState

            t_par = time.perf_counter()
            n_s = 18 + g
            xs = circuits.random_state(n_s, seed=n_s)
            worst, gates_s, remaps_s = 0.0, 0, 0
            for cops in (circuits.h_layer(n_s) + circuits.c2_random_circuit(n_s, 96, seed=28), circuits.c3_qft(n_s)[:120],
                         circuits.c5_grover_iteration(n_s, dense_k3=True)):
                small = DistState(n_s, dist, device, np.complex128, host_staged=dist_backend != "nccl")
                small.upload_global(xs)
                small.apply_ops(cops)
                got = small.download_global()
                remaps_s += small.comm_stats()["remaps"]
                small.close()
                want = O.apply_ops_in_place(n_s, cops, xs.copy())
                worst = max(worst, float(np.max(np.abs(got - want))))
                gates_s += len(cops)
            worst = max_over_ranks(worst)
            st.close()  # (the timed state: its two 2^n_local buffers make room for the checked state and its twin)
            st = None
            quick = args.headline_only or max_over_ranks(0.0 if budget.left() > 240 else 1.0) != 0.0
            parity = W.sharded_parity(lambda: DistState(n, dist, device, np.complex128, host_staged=dist_backend != "nccl"),
                                      dist, n, O, q, circuits, gates=args.gates, quick=quick)
            parity["small_full_vector"] = {"n": n_s, "gates_checked": gates_s, "rows_checked": 3 << n_s, "remaps_exercised": remaps_s,
                                           "max_abs_delta": worst, "ok": bool(worst <= 1e-12)}
            parity["all_legs_ok"] = bool(parity["all_legs_ok"] and worst <= 1e-12)
            parity["seconds"] = round(time.perf_counter() - t_par, 2)
            parity["quick"] = bool(quick)
        except Exception as exc:  # noqa: BLE001
            parity = {"error": repr(exc), "all_legs_ok": False}
        em.detail["parity"] = parity
        res["parity_ok"] = bool(parity.get("all_legs_ok", False))
        res["parity"] = {k: parity.get(k) for k in ("n", "gates_checked", "rows_checked", "max_abs_delta", "bit_equal", "quick", "seconds", "error")
                         if parity.get(k) is not None}
        res["parity"]["small_full_vector_max_abs_delta"] = parity.get("small_full_vector", {}).get("max_abs_delta")
        em.emit(res, "headline + sharded parity")
    if world > 1 and not args.no_extras and res.get("parity_ok") is not False:
        if st is None:  # (the parity block closed the timed state: a fresh one, prepared the same way)
            from rustqip_amd.sharded import DistState

            st = DistState(n, dist, device, np.complex128, host_staged=dist_backend != "nccl")
            st.init_basis(0)
            st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
            sync = st.sync
        run_dist_extras(q, circuits, st, n, args, ops, ops_mixed, em, budget, barrier, max_over_ranks, sync)
    if par is not None:
        em.detail["parity"] = par.detail()
        res["parity"] = par.summary()
        res["parity_ok"] = par.ok()
        par.close()
    em.detail["_now"] = "done"
    if res.get("parity_ok") is False:
        print("bench.py: PARITY FAILED — the measured value is withheld (value: null) and the run exits with status 1; see "
              f"{os.path.basename(DETAIL_PATH)}: parity.legs", file=sys.stderr)
    if not (world == 1 and args.no_extras):
        em.emit(res, "final")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if res.get("parity_ok") is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
