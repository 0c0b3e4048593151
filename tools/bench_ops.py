#!/usr/bin/env python
"""Per-op-class throughput table on one MI355X (not the contract bench; feeds DESIGN.md / profiles/).

  python tools/bench_ops.py [n] > gpurun_out/ops_table.md
Each row: one op type applied REPS times to a resident 2^n Complex<f64> state; GB/s uses the
algorithmic bytes of SURVEY.md §8(d) (qip_hip_op_algorithmic_bytes)."""
import cmath
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402


def _known(fn, key, value):
    """set an option; the measured alternatives exist only in a tuning build of the library (QIP_HIP_TUNING=1 python -m rustqip_amd.build):
    on the product build they are skipped — returns False then, and the row that needed it is left out"""
    try:
        fn(key, value)
        return True
    except q.CircuitError as exc:
        if "unknown" in str(exc):
            return False
        raise

from rustqip_amd import circuits  # noqa: E402


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def hk(k):
    m = np.array([[1.0]])
    for _ in range(k):
        m = np.kron(m, np.array([[1.0, 1.0], [1.0, -1.0]]) / math.sqrt(2.0))
    return m


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "all" else None
    f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
    dtype = np.complex64 if f32 else np.complex128
    amp = 8.0 if f32 else 16.0
    reps = 6
    rng = np.random.default_rng(0)
    hi, mid, lo = 0, n // 2, n - 1  # qubit indices: bit positions n-1, ~n/2, 0
    cases = [
        ("H, target bit n-1", q.make_matrix_op([hi], circuits.H), {}),
        ("H, target bit n/2", q.make_matrix_op([mid], circuits.H), {}),
        ("H, target bit 0 (cross-lane)", q.make_matrix_op([lo], circuits.H), {}),
        ("H, target bit 0 (pair kernel)", q.make_matrix_op([lo], circuits.H), {"lowbit_shuffle": 0}),
        ("X, target bit n/2", q.make_matrix_op([mid], circuits.X), {}),
        ("Rz, target bit n/2", q.make_matrix_op([mid], circuits.rz(0.3)), {}),
        ("Z (phase), bit n/2", q.make_matrix_op([mid], circuits.Z), {}),
        ("T (phase), bit 0", q.make_matrix_op([lo], circuits.T), {}),
        ("CNOT c=n-1 t=n/2", q.make_control_op([hi], q.make_matrix_op([mid], circuits.X)), {}),
        ("CNOT c=0 t=n/2 (control in cache line)", q.make_control_op([lo], q.make_matrix_op([mid], circuits.X)), {}),
        ("CNOT c=n/2 t=0", q.make_control_op([mid], q.make_matrix_op([lo], circuits.X)), {}),
        ("Toffoli", q.make_control_op([hi, 3], q.make_matrix_op([mid], circuits.X)), {}),
        ("controlled-phase", q.make_control_op([hi], q.make_matrix_op([mid], [1, 0, 0, cmath.rect(1, 0.1)])), {}),
        ("Swap(1) bits n-1 <-> 0", q.make_swap_op([hi], [lo]), {}),
        ("Swap(2)", q.make_swap_op([hi, 1], [mid, lo]), {}),
        ("Swap(2), one transposition per sweep", q.make_swap_op([hi, 1], [mid, lo]), {"swap_single": 1}),
        ("Swap(2), 2 groups per lane", q.make_swap_op([hi, 1], [mid, lo]), {"unroll": 2}),
        ("Swap(2) all high bits", q.make_swap_op([hi, 1], [mid, 7]), {}),
        ("Swap(2) all high bits, one transposition per sweep", q.make_swap_op([hi, 1], [mid, 7]), {"swap_single": 1}),
        ("Swap(3)", q.make_swap_op([hi, 1, 2], [mid, lo, lo - 1]), {}),
        ("dense k=2 (VALU regs)", q.make_matrix_op([hi, mid], rand_unitary(2, rng).ravel()), {}),
        ("dense k=2, low bits", q.make_matrix_op([lo, lo - 1], rand_unitary(2, rng).ravel()), {}),
        ("dense k=2, one low bit", q.make_matrix_op([mid, lo], rand_unitary(2, rng).ravel()), {}),
        ("dense k=2, bits 12 and 13", q.make_matrix_op([n - 1 - 12, n - 1 - 13], rand_unitary(2, rng).ravel()), {}),
        ("dense k=3, one low bit", q.make_matrix_op([hi, mid, lo], rand_unitary(3, rng).ravel()), {}),
        ("dense k=3 (MFMA f64)", q.make_matrix_op([hi, mid, 5], rand_unitary(3, rng).ravel()), {}),
        ("dense k=3 (VALU regs)", q.make_matrix_op([hi, mid, 5], rand_unitary(3, rng).ravel()), {"mfma": 0}),
        ("dense k=3 low bits (MFMA f64)", q.make_matrix_op([lo - 2, lo - 1, lo], rand_unitary(3, rng).ravel()), {}),
        ("dense k=3 low bits (VALU regs)", q.make_matrix_op([lo - 2, lo - 1, lo], rand_unitary(3, rng).ravel()), {"mfma": 0}),
        ("controlled dense k=2, low targets, control above the rows", q.make_control_op([3], q.make_matrix_op([lo, lo - 1], rand_unitary(2, rng).ravel())), {}),
        ("controlled dense k=2, high targets", q.make_control_op([3], q.make_matrix_op([hi, mid], rand_unitary(2, rng).ravel())), {}),
        ("controlled dense k=2, control inside a row", q.make_control_op([lo - 3], q.make_matrix_op([hi, lo], rand_unitary(2, rng).ravel())), {}),
        ("2-controlled dense k=3, one low target", q.make_control_op([3, 9], q.make_matrix_op([hi, mid, lo], rand_unitary(3, rng).ravel())), {}),
        ("dense k=4 (MFMA f64)", q.make_matrix_op([hi, mid, 5, lo], rand_unitary(4, rng).ravel()), {}),
        ("dense k=4 (VALU regs)", q.make_matrix_op([hi, mid, 5, lo], rand_unitary(4, rng).ravel()), {"mfma": 0}),
        ("dense k=5 (MFMA f64)", q.make_matrix_op([hi, mid, 5, 7, lo], rand_unitary(5, rng).ravel()), {}),
        ("dense k=5 (MFMA f64, 2 items/iter)", q.make_matrix_op([hi, mid, 5, 7, lo], rand_unitary(5, rng).ravel()), {"unroll": 2}),
        ("dense k=4 high bits (MFMA f64, 1 item/iter)", q.make_matrix_op([hi, mid, 5, 7], rand_unitary(4, rng).ravel()), {"mfma": 2, "unroll": 1}),
        ("dense k=5 high bits (MFMA f64)", q.make_matrix_op([hi, mid, 5, 7, 9], rand_unitary(5, rng).ravel()), {}),
        ("dense k=4 high bits (MFMA f64)", q.make_matrix_op([hi, mid, 5, 7], rand_unitary(4, rng).ravel()), {"mfma": 2}),
        ("dense k=4, two low bits (MFMA f64)", q.make_matrix_op([hi, mid, lo - 1, lo], rand_unitary(4, rng).ravel()), {}),
        ("dense k=4, bits 0-3 (MFMA f64)", q.make_matrix_op([lo - 3, lo - 2, lo - 1, lo], rand_unitary(4, rng).ravel()), {}),
        ("dense k=3 high bits (MFMA f64)", q.make_matrix_op([hi, mid, 5], rand_unitary(3, rng).ravel()), {"mfma": 2}),
        ("dense k=6 (MFMA f64, A streamed)", q.make_matrix_op([hi, mid, 5, 7, lo, 9], rand_unitary(6, rng).ravel()), {}),
        ("dense k=7 (MFMA f64, A streamed)", q.make_matrix_op([hi, mid, 5, 7, lo, 9, 11], rand_unitary(7, rng).ravel()), {}),
        ("dense k=8 (MFMA f64, A streamed)", q.make_matrix_op([hi, mid, 5, 7, lo, 9, 11, 13], rand_unitary(8, rng).ravel()), {}),
        ("dense k=5, real matrix (H on five qubits: two real products)", q.make_matrix_op([hi, mid, 5, 7, lo], hk(5).ravel()), {}),
        ("dense k=6, real matrix (H on six qubits)", q.make_matrix_op([hi, mid, 5, 7, 9, lo], hk(6).ravel()), {}),
        ("dense k=8, real matrix (H on eight qubits: state_bench.rs:118-139's gate)", q.make_matrix_op([hi, mid, 5, 7, 9, 11, 13, lo], hk(8).ravel()), {}),
        ("dense k=9 (MFMA f64, X in LDS, A from L2)", q.make_matrix_op([hi, mid, 5, 7, lo, 9, 11, 13, 17], rand_unitary(9, rng).ravel()), {}),
        ("dense k=10 (MFMA f64, X in LDS in two K phases, A from L2)", q.make_matrix_op([hi, mid, 5, 7, lo, 9, 11, 13, 17, 19], rand_unitary(10, rng).ravel()), {}),
        ("dense k=6 (literal gather)", q.make_matrix_op([hi, mid, 5, 7, lo, 9], rand_unitary(6, rng).ravel()), {"mfma": 0}),
        ("dense k=5 (literal gather)", q.make_matrix_op([hi, mid, 5, 7, lo], rand_unitary(5, rng).ravel()), {"mfma": 0}),
        ("diag k=3 (table)", q.make_matrix_op([hi, mid, lo], np.diag(np.exp(1j * rng.uniform(0, 6, 8))).ravel()), {}),
        ("sparse k=2 (in place)", q.make_sparse_matrix_op([hi, mid], [[(1, 1j)], [(0, 1.0)], [(3, 1.0)], [(2, -1.0)]]), {}),
        ("sparse k=2 (literal gather)", q.make_sparse_matrix_op([hi, mid], [[(1, 1j)], [(0, 1.0)], [(3, 1.0)], [(2, -1.0)]]), {"force_generic": 1}),
        ("sparse k=4, 2 entries per row (in place)", q.make_sparse_matrix_op([hi, mid, 5, 7], [[(r, 0.6), (r ^ 5, 0.8j)] for r in range(16)]), {}),
        ("sparse k=4, 2 entries per row (one group per lane)", q.make_sparse_matrix_op([hi, mid, 5, 7], [[(r, 0.6), (r ^ 5, 0.8j)] for r in range(16)]), {"_sparse_tile": 0}),
        ("sparse k=5, 2 entries per row (in place)", q.make_sparse_matrix_op([hi, mid, 5, 7, 9], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(32)]), {}),
        ("sparse k=5, 2 entries per row (one group per lane)", q.make_sparse_matrix_op([hi, mid, 5, 7, 9], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(32)]), {"_sparse_tile": 0}),
        ("sparse k=16 identity, one entry per row (state_bench.rs:380-393 shape)", q.make_sparse_matrix_op(list(range(16)), [[(r, 1.0)] for r in range(1 << 16)]), {}),
        ("sparse k=16 identity on the low 16 bits", q.make_sparse_matrix_op(list(range(n - 16, n)), [[(r, 1.0)] for r in range(1 << 16)]), {}),
        ("sparse k=8 permutation x phase, scattered bits", q.make_sparse_matrix_op([hi, 3, mid, 7, lo, 11, n - 9, 20], [[(int(c), complex(np.exp(0.1j * r)))] for r, c in enumerate(np.random.default_rng(1).permutation(256))]), {}),
        ("sparse k=6, 2 entries per row", q.make_sparse_matrix_op([hi, mid, 5, 7, 9, 12], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(64)]), {}),
        ("sparse k=8 permutation x phase, scattered bits (out-of-place gather)", q.make_sparse_matrix_op([hi, 3, mid, 7, lo, 11, n - 9, 20], [[(int(c), complex(np.exp(0.1j * r)))] for r, c in enumerate(np.random.default_rng(1).permutation(256))]), {"_sparse_tile": 0}),
        ("sparse k=6, 2 entries per row (out-of-place gather)", q.make_sparse_matrix_op([hi, mid, 5, 7, 9, 12], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(64)]), {"_sparse_tile": 0}),
        ("sparse k=7, 4 entries per row, two positions in the wave row", q.make_sparse_matrix_op([hi, mid, 5, 7, 9, lo, lo - 3], [[(r, 0.5), (r ^ 9, 0.5j), (r ^ 64, -0.5), ((r * 5 + 1) % 128, 0.5)] for r in range(128)]), {}),
        ("controlled sparse k=6, 2 entries per row", q.make_control_op([1], q.make_sparse_matrix_op([hi, mid, 5, 7, 9, 12], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(64)])), {}),
        ("sparse k=6, 2 entries per row (literal gather)", q.make_sparse_matrix_op([hi, mid, 5, 7, 9, 12], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(64)]), {"force_generic": 1}),
        ("H via literal gather", q.make_matrix_op([mid], circuits.H), {"force_generic": 1}),
    ]
    if f32:
        cases = [c for c in cases if "MFMA" not in c[0] and "literal" not in c[0]]
        cases += [("dense k=3 low bits (matrix cores f32)", q.make_matrix_op([lo - 2, lo - 1, lo], rand_unitary(3, rng).ravel()), {}),
                  ("dense k=4, 2 low bits (matrix cores f32)", q.make_matrix_op([hi, mid, lo - 1, lo], rand_unitary(4, rng).ravel()), {}),
                  ("dense k=5 (matrix cores f32)", q.make_matrix_op([hi, mid, 5, 7, lo], rand_unitary(5, rng).ravel()), {}),
                  ("dense k=5 (literal gather)", q.make_matrix_op([hi, mid, 5, 7, lo], rand_unitary(5, rng).ravel()), {"mfma": 0})]
        cases += [("H, target bit n/2 (8-B unpacked path)", q.make_matrix_op([mid], circuits.H), {"packed_f32": 0}),
                  ("Rz, target bit n/2 (8-B unpacked path)", q.make_matrix_op([mid], circuits.rz(0.3)), {"packed_f32": 0})]
    if os.environ.get("QIP_SINGLE_VIA_TILE"):  # tuning aid: 0 = dedicated kernels only, 1 (default) / 2 = one-item tile sweeps
        q.set_global_option("single_via_tile", int(os.environ["QIP_SINGLE_VIA_TILE"]))
    if os.environ.get("QIP_TILE_ROW_SPLIT"):  # 11 (default): split rows in the tile sweeps, 5: contiguous rows
        _known(q.set_global_option, "tile_row_split", int(os.environ["QIP_TILE_ROW_SPLIT"]))
    if os.environ.get("QIP_K4_DIRECT"):
        _known(q.set_global_option, "k4_direct", int(os.environ["QIP_K4_DIRECT"]))
    if os.environ.get("QIP_SINGLE_VIA_TILE_F32"):
        _known(q.set_global_option, "single_via_tile_f32", int(os.environ["QIP_SINGLE_VIA_TILE_F32"]))
    print(f"| op (n={n}, Complex<{'f32' if f32 else 'f64'}>) | kernel | ms | algorithmic GB/s | % of 8 TB/s |\n|---|---|---|---|---|")
    with q.HipState(n, dtype) as st:
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))
        for name, op, opts in cases:
            if only and only not in name:
                continue
            for k in ("lowbit_shuffle", "mfma", "force_generic", "unroll", "packed_f32", "swap_single"):
                _known(st.set_option, k, {"lowbit_shuffle": 1, "mfma": 1, "force_generic": 0, "unroll": 0, "packed_f32": 1, "swap_single": 0}[k])
            ok = _known(q.set_global_option, "sparse_tile", opts.get("_sparse_tile", 1)) or opts.get("_sparse_tile", 1) == 1
            for k, v in opts.items():
                if not k.startswith("_"):
                    ok = _known(st.set_option, k, v) and ok
            if not ok:
                continue  # (a row of a measured alternative: needs a tuning build)
            comp = st.compile_ops([op] * reps)
            st.set_option("profile", 0)
            st.apply_compiled(st.compile_ops([op]))
            st.sync()
            st.set_option("profile", 1)
            st.profile_reset()
            t0 = time.perf_counter()
            st.apply_compiled(comp)
            st.sync()
            dt = (time.perf_counter() - t0) / reps
            prof = st.profile()
            st.profile_reset()
            kern = "+".join(prof) or "-"
            by = q.algorithmic_bytes(n, op, 1 if f32 else 0)
            print(f"| {name} | `{kern}` | {dt*1e3:.3f} | {by/dt/1e9:.0f} | {100*by/dt/1e9/8000:.1f} |")
        st.set_option("profile", 0)
        for k in ("lowbit_shuffle", "mfma", "force_generic", "unroll", "packed_f32", "swap_single"):  # (the last row's options must not leak into the reductions)
            _known(st.set_option, k, {"lowbit_shuffle": 1, "mfma": 1, "force_generic": 0, "unroll": 0, "packed_f32": 1, "swap_single": 0}[k])
        _known(q.set_global_option, "sparse_tile", 1)
        for name, fn, by in (("norm_sqr", st.norm_sqr, amp * 2**n), ("measure_probs k=1", lambda: st.measure_probs([mid]), amp * 2**n),
                             ("measure_probs k=3", lambda: st.measure_probs([hi, mid, lo]), amp * 2**n),
                             ("measure_probs k=12 top bits", lambda: st.measure_probs(list(range(12))), amp * 2**n),
                             ("measure_probs k=8 low bits", lambda: st.measure_probs(list(range(n - 8, n))), amp * 2**n),
                             ("measure_probs k=12 mixed bits", lambda: st.measure_probs([0, n - 1, 3, n - 4, 7, n - 9, 11, n - 13, 15, n - 17, n - 2, 1]), amp * 2**n),
                             ("measure_probs k=16", lambda: st.measure_probs(list(range(2, 18))), amp * 2**n),
                             ("soft_measure (2 passes)", lambda: st.soft_measure([0, mid, lo], 0.4321), amp * 2**n),
                             ("soft_measure (one launch: chunk sums + last block's walk; measured alternative)",
                              lambda: (_known(q.set_global_option, "soft_measure_one_pass", 1), st.soft_measure([0, mid, lo], 0.4321),
                                       _known(q.set_global_option, "soft_measure_one_pass", 0)), amp * 2**n)):
            fn()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            dt = (time.perf_counter() - t0) / 3
            print(f"| {name} | reduction | {dt*1e3:.3f} | {by/dt/1e9:.0f} | {100*by/dt/1e9/8000:.1f} |")
        print(f"\nnorm after all ops: {st.norm_sqr():.15f}")


if __name__ == "__main__":
    main()
