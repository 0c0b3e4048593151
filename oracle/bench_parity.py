"""TEST INFRASTRUCTURE — bench.py's checker and CPU-baseline legs (nothing here is ever the thing measured or shipped).

  Parity        every configuration bench.py TIMES, held against the CPU oracle at the timed size before its number is
                reported: closed sub-cubes against qip_oracle's apply_op_overwrite / apply_op_row (oracle/window_parity.py),
                a twin state that follows gate by gate through the literal kernel and is compared over all 2^n amplitudes
                after every step, closed-form marginals while the state is a product state.  `core()` is the part the
                contract line waits for (the headline and the configs[1] mix, gate by gate); the other modes are checked
                leg by leg right before the extras leg that times them.
  cpu_baseline  the C restatement of qip-iterators apply_op_overwrite timed on the host's cores.
  cpu_shape     the same restatement on one of the reference's own bench shapes.

Only bench.py imports this module.
"""
from __future__ import annotations

import os
import statistics
import time

import numpy as np

from oracle import qip_oracle as O
from oracle import window_parity as W


def host_threads():
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


def cpu_baseline(q, circuits, budget_s: float):
    """Time the oracle on a bounded sample of the headline workload at n = 28: the first gates of the same
    seeded single-qubit circuit, as many as fit the budget (4..16), median of 3 repetitions, all usable cores; plus the same
    loop on ONE thread at n = 22, so that the scaling over threads is visible.  Both buffers are first touched inside the
    OpenMP region (two untimed gates write them in parallel with the static split the timed gates use)."""
    omp_threads, usable, quota, threads = host_threads()

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
    n_gates = int(max(4, min(16, budget_s / (reps * per_gate))))
    ops = circuits.c2_random_circuit(n_cpu, n_gates, seed=28, single_only=True)
    ts = run(n_cpu, ops, reps, threads)
    t = statistics.median(ts)
    by = sum(q.algorithmic_bytes(n_cpu, op) for op in ops)
    n_one = 22  # one thread, a 64x smaller vector, two gates
    ops1 = circuits.c2_random_circuit(n_one, 2, seed=28, single_only=True)
    t1 = statistics.median(run(n_one, ops1, 2, 1))
    O.set_num_threads(omp_threads)
    ns_row_all = 1e9 * t / n_gates / 2 ** n_cpu * threads
    ns_row_one = 1e9 * t1 / len(ops1) / 2 ** n_one
    return {
        "value": by / t / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"first {n_gates} gates of the headline circuit at n={n_cpu} (2 x {16 * 2**n_cpu / 2**30:.0f} GiB), C restatement of "
                  f"qip-iterators 1.5.0 apply_op_overwrite, gcc -O3 -fopenmp, {threads} threads, median of {reps} ({sum(ts):.1f} s CPU work)",
        "gates_per_s": n_gates / t, "ms_per_gate": 1e3 * t / n_gates,
        "detail": {
            "cores_usable": usable, "omp_max_threads": omp_threads, "cgroup_cpu_quota": quota, "reps_s": [round(x, 3) for x in ts],
            "ns_per_row_per_thread": ns_row_all,
            "one_thread": {"n": n_one, "ms_per_gate": 1e3 * t1 / len(ops1), "ns_per_row": ns_row_one, "GBps": 32.0 * 2 ** n_one / (t1 / len(ops1)) / 1e9},
            "thread_scaling_efficiency": ns_row_one / ns_row_all if ns_row_all > 0 else None,
        },
    }


def cpu_shape(n, dtype, op):
    """The CPU restatement on one of the reference's bench shapes: out += op . in, as the reference's bench loop does (apply_op,
    matrix_ops.rs:98-123).  The thread count is passed EXPLICITLY to the restatement (asked for "default" it would pick one
    thread below 2^20 rows, qip_oracle_impl.h, and both columns would be the same measurement)."""
    import ctypes as C

    x = np.zeros(1 << n, dtype=dtype)
    y = np.zeros(1 << n, dtype=dtype)
    cop = op.to_c(O._dt(y))  # (converted once: the descriptor of the 2^16-row sparse op takes longer to build than to apply)
    fn = getattr(O._lib, f"qip_oracle_apply_op_{O._suf(y)}")
    omp_default, _, _, threads = host_threads()
    k = max(1, min(64, int(2 ** (22 - n)))) if n < 22 else 2
    row = {}
    # (the team size is passed explicitly, so the two columns are two measurements at every size — at n = 8 the fork of a
    # `threads`-wide team costs more than the 256-row loop, which is what the first column then shows)
    for nt, label in ((threads, "cpu_restatement_us_per_op"), (1, "cpu_restatement_one_thread_us_per_op")):
        O.set_num_threads(nt)
        call = lambda: fn(n, C.byref(cop), x.ctypes.data, x.size, y.ctypes.data, y.size, 0, 0, 1, nt)  # noqa: E731
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
    return row


OPTION_DEFAULTS = {"tile_auto": 1, "pair_floor": 1, "mfma": 1}  # (every other per-handle option defaults to 0)


class _Sabotaged:
    """TEST HOOK (QIP_BENCH_SABOTAGE_PARITY, tests/test_gpu_d_full_size.py): the CHECKER is made to disagree — the oracle's output
    is perturbed on one row — so that bench.py's failure path (parity_ok false, value null, exit status 1) can be exercised.
    The product is not touched."""

    def __getattr__(self, name):
        return getattr(O, name)

    @staticmethod
    def apply_op_overwrite(m, op, x, out, *a, **kw):
        O.apply_op_overwrite(m, op, x, out, *a, **kw)
        out[1] += 1e-9


class Parity:
    def __init__(self, q, st, n, device=0):
        self.q, self.st, self.n = q, st, n
        self.O = _Sabotaged() if os.environ.get("QIP_BENCH_SABOTAGE_PARITY") else O
        self.legs = {}
        self.t_total = 0.0
        t0 = time.perf_counter()
        self.ops0, self.vecs = W.product_state_ops(n, seed=n)
        st.init_basis(0)
        st.apply_ops(self.ops0)
        self.init_err = 0.0
        for off in (0, (1 << n) // 3, (1 << n) - (1 << 16)):
            got = st.download(off, 1 << 16)
            want = W.product_state_window(n, self.vecs, off, 1 << 16)
            self.init_err = max(self.init_err, float(np.max(np.abs(got - want) / np.abs(want))))
        # the twin needs a second 2^n state (and the relabelled / permutation legs a scratch buffer per state): from n = 32 on
        # one GPU that no longer fits 288 GB, and the whole-vector guard is the closed-form marginals alone
        self.twin = W.Twin(st, lambda: q.HipState(n, np.complex128, device=device)) if n <= 31 else None
        self.had_twin = self.twin is not None
        self.guard = W.ProductGuard(n, self.vecs)
        self.guard.check(st)
        self.headline_prefix = []
        self.t_total += time.perf_counter() - t0

    def leg(self, name, ops, exact, gate_by_gate=False, seed=0, bases=2, max_len=64, state=None, **options):
        """one checked leg: `ops` through the state with `options` set, sub-cubes vs the oracle + the twin over all amplitudes"""
        t0 = time.perf_counter()
        st_, twin_ = (self.st, self.twin) if state is None else (state, None)
        if twin_ is not None:
            twin_.resync()  # (timed legs in between moved the checked state on without the twin)
        for k, v in options.items():
            st_.set_option(k, v)
        try:
            r = W.check_circuit(st_, self.n, ops, self.O, gate_by_gate=gate_by_gate, seed=seed, bases_per_step=bases, twin=twin_, max_len=max_len)
        finally:
            for k in options:
                st_.set_option(k, OPTION_DEFAULTS.get(k, 0))
        r["options"] = options
        r["bar"] = "IEEE-equal" if exact else "1e-12"
        r.setdefault("whole_vector_compares", 0)
        r.setdefault("whole_vector_amplitudes_not_equal", 0)
        r.setdefault("whole_vector_max_abs_delta", 0.0)
        r["ok"] = bool((r["bit_equal"] and r["whole_vector_amplitudes_not_equal"] == 0) if exact
                       else (r["max_abs_delta"] <= 1e-12 and r["whole_vector_max_abs_delta"] <= 1e-12))
        r["ok"] = bool(r["ok"] and r["skipped"] == 0)
        r["seconds"] = round(time.perf_counter() - t0, 2)
        self.t_total += time.perf_counter() - t0
        self.legs[name] = r
        return r["ok"]

    def core(self, ops_headline, ops_mixed):
        """What the contract line waits for: the headline's path (one launch per gate) on the first 32 gates of the timed circuit,
        gate by gate — the state stays a product state, so closed-form marginals follow every gate — and the first 32 gates of the
        configs[1] mix (CNOTs included) the same way."""
        a_ops = ops_headline[:32]
        self.headline_prefix = list(a_ops)
        for k0 in range(0, len(a_ops), 8):
            self.leg("single_qubit_gate_by_gate_%d" % (k0 // 8), a_ops[k0:k0 + 8], True, gate_by_gate=True, seed=11 + k0, bases=4)
            t0 = time.perf_counter()
            for op in a_ops[k0:k0 + 8]:
                self.guard.apply(op)
            self.guard.check(self.st)
            self.t_total += time.perf_counter() - t0
        single = {"gates": 0, "steps": 0, "rows": 0, "row_calls": 0, "windows": 0, "skipped": 0, "whole_vector_compares": 0, "seconds": 0.0}
        for k0 in range(0, len(a_ops), 8):
            r = self.legs.pop("single_qubit_gate_by_gate_%d" % (k0 // 8))
            for key in single:
                single[key] += r[key]
            for key in ("max_abs_delta", "whole_vector_max_abs_delta"):
                single[key] = max(single.get(key, 0.0), r[key])
            single["whole_vector_amplitudes_not_equal"] = single.get("whole_vector_amplitudes_not_equal", 0) + r["whole_vector_amplitudes_not_equal"]
            single["bit_equal"] = single.get("bit_equal", True) and r["bit_equal"]
            single["ok"] = single.get("ok", True) and r["ok"]
        single.update({"bar": "IEEE-equal", "options": {},
                       "product_state_marginals": {"checks": self.guard.checks, "index_sets": self.guard.sets,
                                                   "max_rel_err_vs_closed_form": self.guard.worst_rel}})
        single["ok"] = bool(single["ok"] and self.guard.worst_rel <= 1e-11)
        self.legs["single_qubit_gate_by_gate"] = single
        self.leg("mixed_gate_by_gate", ops_mixed[:32], True, gate_by_gate=True, seed=12, bases=4)

    def record(self, name, got, want, gates=1):
        """a leg checked outside the state machinery (the slice-level calls on real vectors): every element of `got` against the
        oracle's `want`, bar = bit equality"""
        differ = int(np.count_nonzero(got != want))
        self.legs[name] = {"ok": differ == 0, "bar": "IEEE-equal", "max_abs_delta": float(np.max(np.abs(got - want), initial=0.0)) if differ else 0.0,
                           "gates": gates, "rows": int(got.size), "whole_vector_compares": 1, "whole_vector_amplitudes_not_equal": differ}
        return differ == 0

    def reset_state(self):
        """back to the seeded product state advanced by the checked headline gates (the checked circuits entangled it): a
        non-uniform state, which is what gets timed"""
        self.st.init_basis(0)
        self.st.apply_ops(self.ops0 + self.headline_prefix)

    def ok(self):
        return bool(all(r["ok"] for r in self.legs.values()) and self.init_err <= 1e-12)

    def failed_legs(self):
        return [k for k, r in self.legs.items() if not r["ok"]]

    def summary(self):
        """the few numbers the contract line carries"""
        legs = self.legs.values()
        tot = lambda key: sum(r.get(key, 0) for r in legs)  # noqa: E731
        exact = [r for r in legs if r["bar"] == "IEEE-equal"]
        return {
            "checker": "CPU oracle on closed sub-cubes + twin state over all 2^n amplitudes" if self.had_twin else "CPU oracle on closed sub-cubes + closed-form marginals",
            "n": self.n, "legs": len(self.legs), "legs_failed": self.failed_legs(),
            "gates_checked": tot("gates"), "rows_checked": tot("rows"),
            "max_abs_delta_IEEE_legs": max([r["max_abs_delta"] for r in exact] or [0.0]),
            "max_abs_delta_1e-12_legs": max([r["max_abs_delta"] for r in legs if r["bar"] != "IEEE-equal"] or [0.0]),
            "whole_vector_compares": tot("whole_vector_compares"),
            "whole_vector_amplitudes_not_equal_IEEE_legs": sum(r["whole_vector_amplitudes_not_equal"] for r in exact),
            "seconds": round(self.t_total, 1),
        }

    def detail(self):
        legs = self.legs.values()
        tot = lambda key: sum(r.get(key, 0) for r in legs)  # noqa: E731
        d = self.summary()
        d.update({
            "checker": "CPU oracle (oracle/qip_oracle.c apply_op_overwrite + apply_op_row) on closed sub-cubes (tested-only bits resolved "
                       "against the cube's base), oracle/window_parity.py; whole-vector guard: "
                       + ("twin state through the literal kernel compared over all 2^n amplitudes after every step + " if self.had_twin else "")
                       + "closed-form marginals of the product state",
            "state": "seeded product state, pairwise distinct amplitudes (closed form checked: max rel err %.1e)" % self.init_err,
            "whole_vector_guard": "twin state + closed-form marginals" if self.had_twin else "closed-form marginals only (no room for a twin state)",
            "gates_skipped": tot("skipped"), "windows": tot("windows"), "apply_op_row_calls": tot("row_calls"),
            "amplitudes_per_whole_vector_compare": 1 << self.n,
            "all_legs_ok": self.ok(), "legs": self.legs,
        })
        return d

    def close(self):
        if self.twin is not None:
            self.twin.close()
            self.twin = None
