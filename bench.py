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
The configs[1] mix itself (3/4 single-qubit + 1/4 CNOT) is reported beside it as `mixed_circuit`.

Also on the same JSON line:
  parity        the timed configuration checked against the CPU ORACLE before anything is timed: a seeded
                product state with pairwise distinct amplitudes, then gate by gate >= 4 closed sub-cubes of 2^16+
                rows (bottom and top of the index space included) downloaded before/after and compared with the
                oracle's apply_op_overwrite / apply_op_row (oracle/window_parity.py) — the checker, never timed
  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration, measured with
                HIP events on the launching stream inside the timed region
  cpu_baseline  the CPU oracle (C restatement of qip-iterators apply_op_overwrite, OpenMP over all
                host cores) timed on a bounded sample of the same circuit (rank 0, N = 1 only)
  extras        the other BASELINE configs and the optional modes, each the median of 5 repetitions
"""
from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
REPS = 5                # repetitions of every untimed-contract leg (median reported)


def parse_args():
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
                    help="only the timed headline steps (profiling passes: no parity check, mixed circuit, extras or CPU baseline)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--dist-overlap", type=int, default=0,
                    help="N > 1 only: add legs with the exchange cut into this many slices and overlapped with the neighbouring tile sweeps "
                         "(option dist_overlap; off by default: RCCL on two streams has never run on real multi-GPU hardware here)")
    return ap.parse_args()


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


def host_threads(O):
    """(OpenMP's default count, CPUs usable by affinity, the cgroup CPU quota or None, the thread count the CPU legs use = the smallest)"""
    omp_threads = O.max_threads()
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # "max 100000" or "<quota> <period>"
            qv, per = f.read().split()
            quota = None if qv == "max" else float(qv) / float(per)
    except Exception:
        pass
    return omp_threads, usable, quota, max(1, min(omp_threads, usable, int(quota) if quota and quota >= 1 else omp_threads))


def cpu_baseline(q, circuits, args):
    """Time the oracle on a bounded sample of the same workload at n = 28: the first gates of the same
    seeded single-qubit circuit, as many as fit the budget (>= 4), median of 3 repetitions, all usable cores; plus the same
    loop on ONE thread at n = 22, so that the scaling over threads is visible.  Both buffers are first touched inside the
    OpenMP region (two untimed gates write them in parallel with the static split the timed gates use)."""
    import numpy as np

    from oracle import qip_oracle as O

    omp_threads, usable, quota, threads = host_threads(O)

    def run(n, ops, reps, nthreads):
        O.set_num_threads(nthreads)
        state = np.zeros(1 << n, dtype=np.complex128)
        state[0] = 1
        arena = np.zeros_like(state)
        for op in circuits.h_layer(n)[:2]:  # first touch of both buffers, in parallel
            O.apply_op_overwrite(n, op, state, arena)
            state, arena = arena, state
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for op in ops:
                O.apply_op_overwrite(n, op, state, arena)
                state, arena = arena, state
            ts.append(time.perf_counter() - t0)
        return ts

    reps = 3
    n_cal = 22
    t_cal = statistics.median(run(n_cal, circuits.c2_random_circuit(n_cal, 4, seed=28, single_only=True), 2, threads)) / 4
    n_cpu = 28
    per_gate = t_cal * 2 ** (n_cpu - n_cal)
    n_gates = int(max(4, min(16, args.cpu_budget_s / (reps * per_gate))))
    ops = circuits.c2_random_circuit(n_cpu, n_gates, seed=28, single_only=True)
    ts = run(n_cpu, ops, reps, threads)
    t = statistics.median(ts)
    by = circuit_bytes(q, n_cpu, ops)
    # one thread, a 64x smaller vector, two gates
    n_one = 22
    ops1 = circuits.c2_random_circuit(n_one, 2, seed=28, single_only=True)
    t1 = statistics.median(run(n_one, ops1, 2, 1))
    O.set_num_threads(omp_threads)
    ns_row_all = 1e9 * t / n_gates / 2 ** n_cpu * threads
    ns_row_one = 1e9 * t1 / len(ops1) / 2 ** n_one
    return {
        "value": by / t / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
        "cores_usable": usable, "omp_max_threads": omp_threads, "cgroup_cpu_quota": quota,
        "gates_per_s": n_gates / t, "ms_per_gate": 1e3 * t / n_gates, "reps_s": [round(x, 3) for x in ts],
        "ns_per_row_per_thread": ns_row_all,
        "one_thread": {"n": n_one, "ms_per_gate": 1e3 * t1 / len(ops1), "ns_per_row": ns_row_one, "GBps": 32.0 * 2 ** n_one / (t1 / len(ops1)) / 1e9},
        "thread_scaling_efficiency": ns_row_one / ns_row_all if ns_row_all > 0 else None,
        "sample": f"first {n_gates} gates of the same seeded single-qubit circuit at n={n_cpu} (2 buffers x {16 * 2**n_cpu / 2**30:.2f} GiB), "
                  f"C restatement of qip-iterators 1.5.0 apply_op_overwrite, gcc -O3 -fopenmp, {threads} threads "
                  f"({usable} usable cores, cgroup quota {quota}), median of {reps} repetitions ({sum(ts):.1f} s of CPU work); "
                  f"one-thread figure: 2 gates at n={n_one}",
    }


def parity_check(q, circuits, st, n, ops_headline, ops_mixed, gates):
    """Every configuration this bench TIMES against the oracle before anything is timed (see module docstring): the
    headline gate by gate, the mix, the tile sweeps in every mode (interpreted, run-time-compiled, relabelled, tile = 2,
    fused multiply-adds), dense fusion, and the other BASELINE circuits (QFT, Clifford+T, a Grover iteration) through the
    run-time-compiled sweeps at full size.  Two nets: closed sub-cubes against the oracle (rounding-level equality of the
    compared rows) and a whole-vector guard — a twin state that follows gate by gate through the literal kernel and is
    compared over all 2^n amplitudes after every step, plus closed-form marginals while the state is a product state.
    Leaves the state in the seeded product state advanced by the checked gates — a non-uniform state, which is also what
    gets timed."""
    import numpy as np

    from oracle import qip_oracle as O
    from oracle import window_parity as W

    if os.environ.get("QIP_BENCH_SABOTAGE_PARITY"):
        # TEST HOOK (tests/test_parity_gpu.py::test_bench_fails_when_parity_fails): the CHECKER is made to disagree — the
        # oracle's output is perturbed by one ulp-sized nudge on one row — so that the failure path of this script (parity_ok
        # false, value null, exit status 1) can be exercised.  The product is not touched.
        real = O.apply_op_overwrite

        class _Sabotaged:
            def __getattr__(self, name):
                return getattr(O_real, name)

            @staticmethod
            def apply_op_overwrite(m, op, x, out, *a, **kw):
                real(m, op, x, out, *a, **kw)
                out[1] += 1e-9

        O_real, O = O, _Sabotaged()

    t0 = time.perf_counter()
    ops0, vecs = W.product_state_ops(n, seed=n)
    st.init_basis(0)
    st.apply_ops(ops0)
    init_err = 0.0
    for off in (0, (1 << n) // 3, (1 << n) - (1 << 16)):
        got = st.download(off, 1 << 16)
        want = W.product_state_window(n, vecs, off, 1 << 16)
        init_err = max(init_err, float(np.max(np.abs(got - want) / np.abs(want))))
    # the twin needs a second 2^n state (and the relabelled / permutation legs a scratch buffer per state): from n = 32 on one
    # GPU that no longer fits 288 GB, and the whole-vector guard is the closed-form marginals alone (single-qubit legs)
    twin = W.Twin(st, lambda: q.HipState(n, np.complex128, device=st_device(st))) if n <= 31 else None
    big = n >= 33  # one 128-GiB buffer: no leg may take the out-of-place path
    guard = W.ProductGuard(n, vecs)
    guard.check(st)
    legs = {}

    def leg(name, ops, exact, gate_by_gate=False, seed=0, bases=2, max_len=64, state=None, **options):
        st_, twin_ = (st, twin) if state is None else (state, None)
        for k, v in options.items():
            st_.set_option(k, v)
        r = W.check_circuit(st_, n, ops, O, gate_by_gate=gate_by_gate, seed=seed, bases_per_step=bases, twin=twin_, max_len=max_len)
        for k in options:
            st_.set_option(k, 0)
        r["options"] = options
        r["bar"] = "IEEE-equal" if exact else "1e-12"
        r.setdefault("whole_vector_compares", 0)
        r.setdefault("whole_vector_amplitudes_not_equal", 0)
        r.setdefault("whole_vector_max_abs_delta", 0.0)
        r["ok"] = bool((r["bit_equal"] and r["whole_vector_amplitudes_not_equal"] == 0) if exact
                       else (r["max_abs_delta"] <= 1e-12 and r["whole_vector_max_abs_delta"] <= 1e-12))
        r["ok"] = bool(r["ok"] and r["skipped"] == 0)
        if not exact and twin_ is not None:
            twin_.resync()
        legs[name] = r

    # the headline, gate by gate; the state stays a product state: closed-form marginals after every gate
    a_ops = ops_headline[:32]
    for k0 in range(0, len(a_ops), 8):
        leg("single_qubit_gate_by_gate_%d" % (k0 // 8), a_ops[k0:k0 + 8], True, gate_by_gate=True, seed=11 + k0, bases=4)
        for op in a_ops[k0:k0 + 8]:
            guard.apply(op)
        guard.check(st)
    single = {"gates": 0, "steps": 0, "rows": 0, "row_calls": 0, "windows": 0, "skipped": 0, "whole_vector_compares": 0}
    for k0 in range(0, len(a_ops), 8):
        r = legs.pop("single_qubit_gate_by_gate_%d" % (k0 // 8))
        for key in single:
            single[key] += r[key]
        for key in ("max_abs_delta", "whole_vector_max_abs_delta"):
            single[key] = max(single.get(key, 0.0), r[key])
        single["whole_vector_amplitudes_not_equal"] = single.get("whole_vector_amplitudes_not_equal", 0) + r["whole_vector_amplitudes_not_equal"]
        single["bit_equal"] = single.get("bit_equal", True) and r["bit_equal"]
        single["ok"] = single.get("ok", True) and r["ok"]
    single.update({"bar": "IEEE-equal", "product_state_marginals": {"checks": guard.checks, "index_sets": guard.sets,
                                                                    "max_rel_err_vs_closed_form": guard.worst_rel}})
    single["ok"] = bool(single["ok"] and guard.worst_rel <= 1e-11)
    legs["single_qubit_gate_by_gate"] = single
    leg("mixed_gate_by_gate", ops_mixed[:32], True, gate_by_gate=True, seed=12, bases=4)
    leg("mixed_tile1_chunks", ops_mixed[32:96], True, seed=13, bases=4, tile=1)
    leg("mixed_tile1_jit_chunks", ops_mixed[96:160], True, seed=14, tile=1, tile_jit=1)
    if big:
        return finish_parity(q, st, n, legs, twin, ops0, a_ops, init_err, t0)
    # ... with the scheduler relabelling the qubits (tile_relabel = 2: unconditionally, so that every chunk goes through
    # in-tile swaps and the closing bit-permutation sweep); two Swap ops ride along as label exchanges
    swaps = [q.make_swap_op([3], [n - 2]), q.make_swap_op([n - 9], [0])]
    leg("mixed_tile1_jit_relabel_chunks", ops_mixed[160:192] + swaps + ops_mixed[192:224], True, seed=15, tile=1, tile_jit=1, tile_relabel=2)
    # the 1e-12 modes that are timed: commuting reorder (interpreted, compiled, compiled with fused multiply-adds), dense fusion
    more = circuits.c2_random_circuit(n, 192, seed=29)
    leg("mixed_tile2_chunks", more[:48], False, seed=16, tile=2)
    leg("mixed_tile2_jit_chunks", more[48:96], False, seed=17, tile=2, tile_jit=1)
    leg("mixed_tile2_jit_fma_merge_relabel_chunks", more[96:144], False, seed=18, tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_relabel=1)
    leg("mixed_fuse5_chunks", more[144:192], False, seed=19, fuse=5)
    # the other BASELINE circuits as they are timed: run-time-compiled sweeps at full size.  QFT's controlled phases only
    # TEST their bits, so a chunk is closed over its H targets alone and holds what a timed segment holds.
    leg("configs2_qft_tile1_jit", circuits.c3_qft(n), True, seed=20, max_len=160, tile=1, tile_jit=1)
    # (Clifford+T and Grover: the first half of what is timed — every gate kind and the 29-control Z included — keeps the block ~1 min)
    leg("configs3_clifford_t_tile1_jit", circuits.c4_clifford_t(n, gates, seed=32)[:gates // 2], True, seed=21, tile=1, tile_jit=1)
    leg("configs4_grover_tile1_jit", circuits.c5_grover_iteration(n)[:100], True, seed=22, max_len=96, tile=1, tile_jit=1)
    leg("configs4_grover_dense_k3_tile1_jit", circuits.c5_grover_iteration(n, dense_k3=True)[:80], False, seed=23, max_len=96, tile=1, tile_jit=1)
    leg("configs2_qft_tile2_jit_fma_merge", circuits.c3_qft(n)[:200], False, seed=24, max_len=160, tile=2, tile_jit=1, tile_fma=1, tile_merge=1)
    leg("configs3_clifford_t_tile2_jit_fma_merge_relabel", circuits.c4_clifford_t(n, gates, seed=32)[gates // 2:gates // 2 + 64], False, seed=25,
        tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_relabel=1)
    # r4: wide tiles (13-bit register-resident tile, seven free positions per sweep, run-time-compiled) as they are timed
    morew = circuits.c2_random_circuit(n, 128, seed=30)
    leg("mixed_tile1_jit_wide_chunks", morew[:64], True, seed=26, tile=1, tile_jit=1, tile_wide=1)
    leg("mixed_tile1_jit_wide_relabel_chunks", morew[64:128], True, seed=27, tile=1, tile_jit=1, tile_wide=1, tile_relabel=2)
    leg("configs3_clifford_t_tile1_jit_wide_relabel", circuits.c4_clifford_t(n, gates, seed=32)[gates // 2 + 64:gates], True, seed=28,
        tile=1, tile_jit=1, tile_wide=1, tile_relabel=1)
    leg("configs4_grover_tile1_jit_wide", circuits.c5_grover_iteration(n)[100:], True, seed=29, max_len=96, tile=1, tile_jit=1, tile_wide=1)
    # r5: the dense-k3 variant on wide tiles is a timed leg now (its 8 x 8 gates written out group by group): the rest of the iteration
    leg("configs4_grover_dense_k3_tile1_jit_wide", circuits.c5_grover_iteration(n, dense_k3=True)[80:], False, seed=34, max_len=96, tile=1, tile_jit=1, tile_wide=1)
    more2 = circuits.c2_random_circuit(n, 64, seed=31)
    leg("mixed_tile2_jit_fma_merge_wide_chunks", more2, False, seed=30, tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1)
    leg("configs2_qft_tile2_jit_fma_merge_wide", circuits.c3_qft(n)[200:400], False, seed=31, max_len=160, tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1)
    leg("configs3_clifford_t_tile2_jit_fma_merge_wide_relabel", circuits.c4_clifford_t(n, gates, seed=33)[:64], False, seed=32,
        tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1, tile_relabel=1)
    if twin is not None:  # (closed before the f32 state below is created: HBM holds st + its scratch + the f32 pair)
        twin.close()
        twin = None
    # r5: the Complex<f32> leg that is timed on wide tiles (extras.complex64_n*.mixed_tile1_jit_wide) — a Complex<f32>
    # state of the timed size against the f32 ORACLE on closed sub-cubes; both sides compute in unfused f32: bit equality
    with q.HipState(n, np.complex64, device=st_device(st)) as s32:
        s32.init_basis(0)
        s32.apply_ops(ops0)
        leg("complex64_mixed_tile1_jit_wide_chunks", ops_mixed[:64], True, seed=33, state=s32, tile=1, tile_jit=1, tile_wide=1)
    return finish_parity(q, st, n, legs, twin, ops0, a_ops, init_err, t0)


def finish_parity(q, st, n, legs, twin, ops0, a_ops, init_err, t0):
    if twin is not None:
        twin.close()
    # back to a product state for the timed part (the checked circuits entangled it): re-prepare and advance as before
    st.init_basis(0)
    st.apply_ops(ops0 + a_ops)
    tot = lambda key: sum(r.get(key, 0) for r in legs.values())  # noqa: E731
    exact_legs = [r for r in legs.values() if r["bar"] == "IEEE-equal"]
    return {
        "checker": "CPU oracle (oracle/qip_oracle.c apply_op_overwrite + apply_op_row) on closed sub-cubes (tested-only bits resolved "
                   "against the cube's base), oracle/window_parity.py; whole-vector guard: twin state through the literal kernel compared "
                   "over all 2^n amplitudes after every step + closed-form marginals of the product state",
        "n": n, "state": "seeded product state, pairwise distinct amplitudes (closed form checked: max rel err %.1e)" % init_err,
        "gates_checked": tot("gates"), "gates_skipped": tot("skipped"), "rows_checked": tot("rows"), "windows": tot("windows"),
        "apply_op_row_calls": tot("row_calls"),
        "max_abs_delta": max(r["max_abs_delta"] for r in exact_legs),
        "bit_equal": bool(all(r["bit_equal"] for r in exact_legs)),
        "max_abs_delta_1e-12_legs": max([r["max_abs_delta"] for r in legs.values() if r["bar"] != "IEEE-equal"] or [0.0]),
        "whole_vector_guard": "twin state + closed-form marginals" if twin is not None else "closed-form marginals only (no room for a twin state)",
        "whole_vector": {"compares": tot("whole_vector_compares"), "amplitudes_per_compare": 1 << n,
                         "amplitudes_not_equal_in_IEEE_legs": sum(r["whole_vector_amplitudes_not_equal"] for r in exact_legs),
                         "max_abs_delta_all_legs": max(r["whole_vector_max_abs_delta"] for r in legs.values())},
        "all_legs_ok": bool(all(r["ok"] for r in legs.values()) and init_err <= 1e-12),
        "legs": legs,
        "seconds": round(time.perf_counter() - t0, 2),
    }


def st_device(st):
    return getattr(st, "device", 0)


def reference_bench_shapes(q, circuits, args):
    """The reference's OWN benches (the only numbers its code defines: `cargo +nightly bench`), at the reference's sizes:
    qip/benches/state_bench.rs:118-139 (n = 8, dense 8-qubit H^8), :141-155 (n = 24, H on qubit 0), :157-170 (n = 8, 7-control
    identity), :172-202 (n = 16, 15-control identity, Complex<f64> and Complex<f32>), :380-393 (n = 16, 16-qubit sparse identity);
    qip-iterators/benches/matmul_bench.rs:163-177 (n = 20, [1,1,1,1] on qubit 0; real f64 there, complex here).  One `apply_op`
    per iteration there; here microseconds per op three ways — eager (one C-ABI call per op, one sync at the end), a hipGraph
    program of 64 ops, the same program over tile sweeps (tile = 1) — on a resident state, beside the CPU restatement's
    apply_op (accumulate, like the reference's bench loop) on the host's cores.  These states live in L2 / MALL: launch-bound."""
    import numpy as np

    from oracle import qip_oracle as O

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
                one = s.compile_ops([op])
                s.apply_compiled(one)
                s.sync()
                ts = []
                for _ in range(REPS):
                    s.sync()
                    t = time.perf_counter()
                    for _ in range(reps):
                        s.apply_compiled(one)
                    s.sync()
                    ts.append((time.perf_counter() - t) / reps)
                row["eager_us_per_op"] = 1e6 * statistics.median(ts)
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
        if not args.no_cpu_baseline:
            # the CPU restatement on the same shape: out += op . in, as the reference's bench loop does (apply_op, matrix_ops.rs:98-123)
            try:
                import ctypes as C

                x = np.zeros(1 << n, dtype=dtype)
                y = np.zeros(1 << n, dtype=dtype)
                cop = op.to_c(O._dt(y))  # (converted once: the descriptor of the 2^16-row sparse op takes longer to build than to apply)
                fn = getattr(O._lib, f"qip_oracle_apply_op_{O._suf(y)}")
                call = lambda: fn(n, C.byref(cop), x.ctypes.data, x.size, y.ctypes.data, y.size, 0, 0, 1, 0)  # noqa: E731
                omp_default, _, _, threads = host_threads(O)
                k = max(1, min(64, int(2 ** (22 - n)))) if n < 22 else 2
                for label, nt in (("cpu_restatement_us_per_op", threads), ("cpu_restatement_one_thread_us_per_op", 1)):
                    O.set_num_threads(nt)  # (the cgroup's CPU quota, not the 128+ threads OpenMP would start; and one thread: at n = 8 the fork costs more than the work)
                    call()
                    ts = []
                    for _ in range(3):
                        t = time.perf_counter()
                        for _ in range(k):
                            call()
                        ts.append((time.perf_counter() - t) / k)
                    row[label] = 1e6 * statistics.median(ts)
                O.set_num_threads(omp_default)
                row["cpu_threads"] = threads
            except Exception as exc:  # noqa: BLE001
                row["cpu_error"] = repr(exc)
        out[name] = row
    return out


def main():
    args = parse_args()
    if args.headline_only:
        args.no_parity = args.no_extras = args.no_cpu_baseline = True
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
        import torch.distributed as dist

        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=dist_backend)

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

    parity = None
    if world == 1:
        st = q.HipState(n, np.complex128, device=device)
        if not args.no_parity:
            parity = parity_check(q, circuits, st, n, ops, ops_mixed, args.gates)
        else:  # same resident state as the checked run (seeded product state), without the oracle comparison
            from oracle import window_parity as W

            st.init_basis(0)
            st.apply_ops(W.product_state_ops(n, seed=n)[0])
        compiled = st.compile_ops(ops)
        run_step = lambda: st.apply_compiled(compiled)
        sync = st.sync
        set_profile = lambda v: st.set_option("profile", v)
        get_profile = lambda: (st.profile(), st.profile_reset())[0]
    else:
        from rustqip_amd.sharded import DistState

        # the sharded state inside libqip_hip.so (C ABI qip_hip_dist_*): planner, pack sweep and the RCCL exchange
        st = DistState(n, dist, device, np.complex128, host_staged=dist_backend != "nccl")
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
        st.comm_stats()  # reset: the headline's comm figures cover warm-up + timed steps only
        compiled = st.compile_ops(ops)
        run_step = lambda: st.apply_compiled(compiled)
        sync = st.sync
        set_profile = st.set_profile
        get_profile = st.take_profile

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

    ms_per_step = 1e3 * elapsed / args.steps
    value = bytes_per_step * args.steps / elapsed / 1e9
    kname, kstat = dominant_kernel(profile)
    roofline = None
    if kstat:
        per_launch_bytes = kstat["algorithmic_bytes"] / kstat["launches"]
        avg_ms = kstat["total_ms"] / kstat["launches"]
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = load_traffic(kname)
        if args.n_local != 30 and traffic is not None:
            # the PMC passes were taken at the default size (2^30 amplitudes per GPU): the traffic of a sweep is proportional
            # to the shard (the kernels are the same), so the per-launch figure is scaled, and labelled as such
            traffic *= 2.0 ** (args.n_local - 30)
            traffic_src = f"{traffic_src}, measured at 2^30 amplitudes and scaled to 2^{args.n_local}"
        if traffic_stale:
            print("bench.py: warning: profiles/pmc_traffic.json was measured on an older csrc/qip_kernels.h — roofline.traffic is "
                  "flagged stale; re-run tools/profile_round.sh", file=sys.stderr)
        roofline = {
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_source": f"static, from {traffic_src} (separate rocprofv3 --pmc passes of this command; not re-measured per run)",
            "traffic_stale": traffic_stale,
            "avg_launch_ms": avg_ms, "launches": kstat["launches"], "algorithmic_bytes_per_launch": per_launch_bytes,
        }
    kernels = {
        k: {"launches": v["launches"], "avg_ms": v["total_ms"] / v["launches"],
            "GBps": (v["algorithmic_bytes"] / v["launches"]) / (v["total_ms"] / v["launches"] * 1e-3) / 1e9}
        for k, v in (profile or {}).items() if v["launches"] and v["total_ms"] > 0
    }

    def leg(cops, label_ops="gates", **options):
        """median of REPS timed applications of a circuit with the given state options (single GPU)"""
        for k, v in options.items():
            st.set_option(k, v)
        cc = st.compile_ops(cops)
        st.set_option("profile", 1)
        st.apply_compiled(cc)
        st.sync()
        st.profile_reset()
        dt, ts = median_time(lambda: st.apply_compiled(cc), st.sync)
        prof = st.profile()
        st.set_option("profile", 0)
        for k in options:
            st.set_option(k, 0)
        sweeps = sum(v["launches"] for v in prof.values()) // (REPS + 1)
        sweep_bytes = sum(v["algorithmic_bytes"] for v in prof.values()) / (REPS + 1)
        by = circuit_bytes(q, n, cops)
        return {label_ops: len(cops), "ms": 1e3 * dt, "ms_min_max": [round(1e3 * min(ts), 3), round(1e3 * max(ts), 3)],
                "%s_per_s" % label_ops: len(cops) / dt, "algorithmic_GBps": by / dt / 1e9, "launches": sweeps,
                "per_launch_GBps": sweep_bytes / dt / 1e9, "reps": REPS}

    extras = {}
    mixed = None
    if world == 1 and not args.headline_only:
        mixed = leg(ops_mixed)
        mixed["frac_of_8TBps"] = mixed["algorithmic_GBps"] / HBM_PEAK_GBPS
        mixed["workload"] = "configs[1] generator: 3/4 H/X/Rz + 1/4 CNOT, seed 28, %d gates, n=%d" % (len(ops_mixed), n)
    if world == 1 and not args.no_extras:
        # per-target-qubit H sweep at n_local: GB/s by target qubit, median of REPS x 2 gates
        sweep = []
        for tq in range(n):
            op = st.compile_ops([q.make_matrix_op([tq], circuits.H)] * 2)
            dt, _ = median_time(lambda: st.apply_compiled(op), st.sync)
            sweep.append(round(32.0 * 2**n / (dt / 2) / 1e9, 1))
        extras["h_sweep_GBps_by_target_qubit"] = sweep
        extras["h_sweep_min_median_GBps"] = [min(sweep), float(np.median(sweep))]
        # optional modes on the configs[1] mix: dense fusion (one sweep per fused gate: per_launch_GBps is per SWEEP
        # bytes, never per-gate bytes over sweep time) and LDS-resident multi-gate sweeps
        extras["fused_k5"] = leg(ops_mixed, fuse=5)
        extras["tiled_mode1"] = leg(ops_mixed, tile=1)
        # the same IEEE-equal sweeps with every segment compiled at run time for that segment (hiprtc; cached): the
        # first application pays the compilation (reported), the timed repetitions replay cached kernels
        import ctypes as _C

        from rustqip_amd import _ffi as _F

        def jit_stats():
            k, ms = _C.c_uint64(), _C.c_double()
            _F.lib.qip_hip_jit_stats(_C.byref(k), _C.byref(ms))
            return int(k.value), ms.value

        k0, ms0 = jit_stats()
        extras["tiled_mode1_jit"] = leg(ops_mixed, tile=1, tile_jit=1)
        k1, ms1 = jit_stats()
        extras["tiled_mode1_jit"].update({"segments_compiled": k1 - k0, "compile_ms_once": ms1 - ms0})
        # ... with the scheduler relabelling the qubits (soonest-needed qubits on index bits 0..5, one closing bit-permutation
        # sweep; only moves are added: still IEEE-equal), and the reordering mode (1e-12 bar) compiled the same way
        extras["tiled_mode1_jit_relabel"] = leg(ops_mixed, tile=1, tile_jit=1, tile_relabel=1)
        # r3: multiply-add contraction + merged runs of diagonal gates in the compiled tile = 2 segments (1e-12 bar); tile_jit = 1
        # compiles a segment's structure and takes its numbers as kernel data (tile_jit = 3: numbers as literals, for comparison)
        extras["tiled_mode2_jit_fma_relabel"] = leg(ops_mixed, tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_relabel=1)
        # r4: wide tiles — a 13-bit tile held in registers (32 amplitudes per lane), seven free positions per sweep, LDS as a
        # transposition buffer; run-time-compiled segments, IEEE-equal in circuit order like the 11-bit sweeps
        extras["tiled_mode1_jit_wide"] = leg(ops_mixed, tile=1, tile_jit=1, tile_wide=1)
        extras["tiled_mode1_jit_wide_relabel"] = leg(ops_mixed, tile=1, tile_jit=1, tile_wide=1, tile_relabel=1)
        # ... and the 1e-12 mode over wide tiles, with fused multiply-adds and merged runs of diagonal gates (without relabelling:
        # seven positions per sweep leave little for it to win, and its in-tile swaps and closing sweep cost more than they save)
        extras["tiled_mode2_jit_fma_merge_wide"] = leg(ops_mixed, tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1)
        # r5: a PROGRAM created on a tile = 1 state (option tile_auto): compiled once at creation, replayed as one hipGraph
        try:
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
        # the other BASELINE configs on the same resident state size
        for cname, cops in (("configs2_qft_n%d" % n, circuits.c3_qft(n)),
                            ("configs3_clifford_t_n%d" % n, circuits.c4_clifford_t(n, args.gates, seed=32)),
                            ("configs4_grover_iteration_n%d" % n, circuits.c5_grover_iteration(n)),
                            ("configs4_grover_dense_k3_n%d" % n, circuits.c5_grover_iteration(n, dense_k3=True))):
            extras[cname] = leg(cops, "ops")
            extras[cname]["tile1"] = leg(cops, "ops", tile=1)
            k0, ms0 = jit_stats()
            extras[cname]["tile1_jit"] = leg(cops, "ops", tile=1, tile_jit=1)
            k1, ms1 = jit_stats()
            extras[cname]["tile1_jit"].update({"segments_compiled": k1 - k0, "compile_ms_once": ms1 - ms0})
            if "qft" not in cname:  # r4: wide tiles (the issue-bound QFT gains nothing; r5: the dense-k3 variant does, its 8 x 8 gates are written out)
                extras[cname]["tile1_jit_wide"] = leg(cops, "ops", tile=1, tile_jit=1, tile_wide=1)
            if "clifford" in cname:  # (QFT and Grover are layered: the scheduler keeps the plain plan for them)
                extras[cname]["tile1_jit_wide_relabel"] = leg(cops, "ops", tile=1, tile_jit=1, tile_wide=1, tile_relabel=1)
                extras[cname]["tile1_jit_relabel"] = leg(cops, "ops", tile=1, tile_jit=1, tile_relabel=1)
                # ... and the 1e-12 mode as it is timed for configs[1] (fused multiply-adds, merged diagonal runs, relabelled)
                extras[cname]["tile2_jit_fma_merge_relabel"] = leg(cops, "ops", tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_relabel=1)
                extras[cname]["tile2_jit_fma_merge_wide_relabel"] = leg(cops, "ops", tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1, tile_relabel=1)
            if "qft" in cname:  # the issue-bound circuit: the 1e-12 mode with fused multiply-adds
                extras[cname]["tile2_jit_fma_merge"] = leg(cops, "ops", tile=2, tile_jit=1, tile_fma=1, tile_merge=1)
                extras[cname]["tile2_jit_fma_merge_wide"] = leg(cops, "ops", tile=2, tile_jit=1, tile_fma=1, tile_merge=1, tile_wide=1)
        extras["norm_sqr_end"] = st.norm_sqr()
        st.close()
        # r5: where the run-time-compiled segments of this process came from (helper processes side by side, disk cache)
        extras["jit"] = dict(_F.jit_counters(), cache_dir=_F.lib.qip_hip_jit_cache_dir().decode())
        extras["reference_bench_shapes"] = reference_bench_shapes(q, circuits, args)
        # configs[1] exactly: n = 28
        n28 = 28
        ops28 = circuits.c2_random_circuit(n28, args.gates, seed=28)
        with q.HipState(n28) as s28:
            s28.init_basis(0)
            s28.apply_ops(circuits.h_layer(n28) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n28)])
            c28 = s28.compile_ops(ops28)
            dt, ts = median_time(lambda: s28.apply_compiled(c28), s28.sync)
            extras["configs1_n28"] = {"GBps": circuit_bytes(q, n28, ops28) / dt / 1e9, "gates_per_s": len(ops28) / dt,
                                      "ms_per_step": 1e3 * dt, "reps": REPS, "norm_sqr": s28.norm_sqr()}

        # the same headline circuit on a Complex<f32> state (8 GiB at n = 30; 16 * 2^n bytes per gate),
        # gate by gate and as tile sweeps (run-time-compiled segments, qubits relabelled); medians of REPS
        with q.HipState(n, np.complex64) as s32:
            s32.init_basis(0)
            s32.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
            c32 = s32.compile_ops(ops)
            dt, _ = median_time(lambda: s32.apply_compiled(c32), s32.sync)
            by32 = sum(q.algorithmic_bytes(n, op, 1) for op in ops)
            f32 = {"gates": len(ops), "ms": 1e3 * dt, "gates_per_s": len(ops) / dt, "algorithmic_GBps": by32 / dt / 1e9,
                   "frac_of_8TBps": by32 / dt / 1e9 / HBM_PEAK_GBPS, "reps": REPS}
            cm32 = s32.compile_ops(ops_mixed)
            for k, v in (("tile", 1), ("tile_jit", 1), ("tile_relabel", 1)):
                s32.set_option(k, v)
            dt, _ = median_time(lambda: s32.apply_compiled(cm32), s32.sync)
            f32["mixed_tile1_jit_relabel"] = {"gates": len(ops_mixed), "ms": 1e3 * dt, "gates_per_s": len(ops_mixed) / dt, "reps": REPS}
            s32.set_option("tile_relabel", 0)
            s32.set_option("tile_wide", 1)  # r4: wide tiles (32 amplitudes per lane are 64 registers in f32)
            dt, _ = median_time(lambda: s32.apply_compiled(cm32), s32.sync)
            f32["mixed_tile1_jit_wide"] = {"gates": len(ops_mixed), "ms": 1e3 * dt, "gates_per_s": len(ops_mixed) / dt, "reps": REPS}
            f32["norm_sqr"] = s32.norm_sqr()
            extras["complex64_n%d" % n] = f32

    if world > 1 and not args.no_extras:
        # BASELINE configs[3] (Clifford+T) and configs[4] (Grover iteration, plain and dense k = 3) on the sharded
        # state, and the headline circuit with the local runs between remaps applied as tile sweeps (tile = 1:
        # IEEE-equal).  Guarded: nothing here can take the bench line down.  Median of REPS, max over ranks.
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
            cs = st.comm_stats()
            return {"ops": len(cops), "ms": 1e3 * dt, "ops_per_s": len(cops) / dt, "algorithmic_GBps": circuit_bytes(q, n, cops) / dt / 1e9,
                    "reps": REPS, "comm_over_reps": cs}

        for cname, cops, kw in (("configs3_clifford_t_n%d" % n, circuits.c4_clifford_t(n, args.gates, seed=32), {}),
                                ("configs4_grover_iteration_n%d" % n, circuits.c5_grover_iteration(n), {}),
                                ("configs4_grover_dense_k3_n%d" % n, circuits.c5_grover_iteration(n, dense_k3=True), {}),
                                ("configs1_mixed_n%d" % n, ops_mixed, {}),
                                ("configs3_clifford_t_tiled_mode1", circuits.c4_clifford_t(n, args.gates, seed=32), {"tile": 1}),
                                ("configs1_mixed_tiled_mode1", ops_mixed, {"tile": 1}),
                                ("headline_tiled_mode1", ops, {"tile": 1}),
                                # r5: the compiled sweeps (wide tiles; the remap's gather rides in their store) ...
                                ("configs1_mixed_tiled_mode1_jit_wide", ops_mixed, {"tile": 1, "jit": 1, "wide": 1}),
                                ("configs3_clifford_t_tiled_mode1_jit_wide", circuits.c4_clifford_t(n, args.gates, seed=32), {"tile": 1, "jit": 1, "wide": 1})) + (
                                # ... and, on request, with the exchange overlapped with the sweeps either side of it
                                (("configs1_mixed_tiled_mode1_jit_wide_overlap", ops_mixed, {"tile": 1, "jit": 1, "wide": 1, "overlap": args.dist_overlap}),
                                 ("configs3_clifford_t_tiled_mode1_jit_wide_overlap", circuits.c4_clifford_t(n, args.gates, seed=32),
                                  {"tile": 1, "jit": 1, "wide": 1, "overlap": args.dist_overlap}),
                                 ("configs1_mixed_tiled_mode1_overlap", ops_mixed, {"tile": 1, "overlap": args.dist_overlap}))
                                if args.dist_overlap >= 2 else ()):
            try:
                extras[cname] = dist_leg(cops, **kw)
            except Exception as exc:  # noqa: BLE001
                extras[cname] = {"error": repr(exc)}
        try:
            extras["norm_sqr_end"] = st.norm_sqr()
        except Exception as exc:  # noqa: BLE001
            extras["norm_sqr_end"] = repr(exc)

    dist_desc = st.describe() if world > 1 else None
    if world > 1 and not args.no_parity:
        # The N > 1 path against the CPU oracle on THIS fabric (the real transport, every rank's real kernels), at the size
        # that was just timed: closed sub-cubes of the logical index space gathered through the layout, a twin sharded state
        # on the literal kernel compared over all 2^n amplitudes after every step, closed-form marginals of the product state
        # (oracle/window_parity.sharded_parity) — plus a small sharded state compared as a gathered full vector.  The
        # checker, never timed.  A failure here is fatal for the line (parity_ok false, value null, rc 1).
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
            parity = W.sharded_parity(lambda: DistState(n, dist, device, np.complex128, host_staged=dist_backend != "nccl"),
                                      dist, n, O, q, circuits, gates=args.gates)
            parity["small_full_vector"] = {"n": n_s, "gates_checked": gates_s, "rows_checked": 3 << n_s, "remaps_exercised": remaps_s,
                                           "max_abs_delta": worst, "ok": bool(worst <= 1e-12)}
            parity["all_legs_ok"] = bool(parity["all_legs_ok"] and worst <= 1e-12)
            parity["seconds"] = round(time.perf_counter() - t_par, 2)
            st = None
        except Exception as exc:  # noqa: BLE001
            parity = {"error": repr(exc), "all_legs_ok": False}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(q, circuits, args)

    # parity is the first gate: a line whose checked legs did not all pass carries no value and the run fails
    parity_ok = None if parity is None else bool(parity.get("all_legs_ok", False))
    if rank == 0:
        line = {
            "metric": "single-qubit gate apply GB/s (algorithmic bytes: 32 * 2^n per H / X / Rz gate)",
            "value": value if parity_ok is not False else None, "unit": "GB/s", "parity_ok": parity_ok, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.gates} random single-qubit gates (uniform H / X / Rz(theta), uniform target; configs[1] generator, seed 28, "
                            f"single-qubit part) at n={n} ({args.n_local} qubits = {16 * 2**args.n_local / 2**30:.0f} GiB per GPU), Complex<f64>, "
                            f"resident non-uniform state",
                "n_qubits": n, "n_local": args.n_local, "gates_per_step": args.gates,
                "algorithmic_bytes_per_step": bytes_per_step,
                "parallelism": "single GPU" if world == 1 else f"state sharded by top {g} index bits over {world} GPUs, RCCL all-to-all qubit remap",
            },
            "gates_per_s": args.gates * args.steps / elapsed,
            "frac_of_hbm_peak_per_gpu": value / world / HBM_PEAK_GBPS,
            "norm_sqr_after": norm,
            "parity": parity,
            "parity_rows_checked": parity.get("rows_checked") if parity else None,
            "max_abs_delta": parity.get("max_abs_delta") if parity else None,
            "roofline": roofline,
            "kernels": kernels,
            "mixed_circuit": mixed,
            "cpu_baseline": cpu,
        }
        if extras:
            line["extras"] = extras
        if world > 1:
            line["comm"] = comm_headline
            line["dist"] = dist_desc
            # what RCCL itself reports (ncclCommCount read back from the communicator, min over ranks): proof that the
            # collective library saw `world` ranks; 0 with the host-staged test transport
            line["rccl_ranks"] = rccl_ranks_min
            ref = load_n1_reference(args.n_local)
            if ref:
                # (aggregate GB/s / G) / single-GPU GB/s at the same shard size, communication included
                line["per_gpu_efficiency"] = value / world / ref["value"]
                line["per_gpu_efficiency_reference"] = ref
        if parity_ok is False:
            line["value_withheld"] = value
            print("bench.py: PARITY FAILED — the measured value is withheld (value: null) and the run exits with status 1; see parity.legs",
                  file=sys.stderr)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity_ok is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
