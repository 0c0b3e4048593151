"""Shared helpers of the GPU parity tests (tests/test_gpu_*.py, one file per SURVEY.md §8 row) — needs a real MI355X (`-m gpu`).

Everything goes through the C ABI (ctypes -> libqip_hip.so -> HIP kernels).  Bars:
  * permutation ops (X, CNOT, SWAP, 0/1 matrices): IEEE `==` on every component;
  * everything else: |delta| <= 1e-12 per amplitude (f64), 1e-5 (f32) — and, because kernels and
    oracle are both built without FMA contraction and fold in the same order, the 1-qubit,
    phase, diagonal and literal-gather kernels are additionally expected to be bit-equal,
    which is asserted where it has been observed.
"""
import cmath
import math
import os
import zlib

import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd import circuits
from rustqip_amd.ops import MatrixOp


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL64 = 1e-12
TOL32 = 1e-5

S2 = math.sqrt(0.5)
GATES_1Q = {
    "X": [0, 1, 1, 0],
    "Y": [0, -1j, 1j, 0],
    "Z": [1, 0, 0, -1],
    "H": [S2, S2, S2, -complex(S2, 0.0)],
    "S": [1, 0, 0, 1j],
    "T": [1, 0, 0, cmath.rect(1, math.pi / 4)],
    "Rz": [cmath.rect(1, -0.35), 0, 0, cmath.rect(1, 0.35)],
    "upper": [1, 1, 0, 1],       # zero entry in a dense matrix (zero-skipping path)
    "rank1": [0.5, 0.25j, 0, 0],  # a zero row
    "ident": [1, 0, 0, 1],
    "dense": [0.3 + 0.1j, -0.7j, 0.2, 0.9 - 0.4j],
}
PERMUTATIONS = {"X", "ident"}
WIDE_DENSE3_INLINE_DEFAULT = 1  # the library's default of global option tile_wide_dense3_inline (restored after tests that flip it)



from conftest import has_tuning_options, needs_tuning  # noqa: E402,F401

_TUNING = None
TUNING_STATE_KEYS = {"lowbit_shuffle", "packed_f32", "tile_passes", "unroll", "swap_single"}


def tuning() -> bool:
    """the library is a -DQIP_HIP_TUNING build: the measured alternatives are options (conftest.has_tuning_options)"""
    global _TUNING
    if _TUNING is None:
        _TUNING = has_tuning_options()
    return _TUNING


def variants(*option_sets):
    """the option sets a test runs an op under; those that need a tuning build are dropped on the product build"""
    return tuple(o for o in option_sets if tuning() or not (set(o) & TUNING_STATE_KEYS))


def rand_state(n, seed, dtype=np.complex128):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return (v / np.linalg.norm(v)).astype(dtype)


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def hip_apply(n, op, x, **options):
    """op . x through a fresh handle with the given per-handle options (a measured alternative that only a tuning build knows
    is dropped on the product build: the call then runs the default path once more)"""
    options = {k: v for k, v in options.items() if k not in TUNING_STATE_KEYS or tuning()}
    with q.HipState(n, x.dtype) as st:
        for k, v in options.items():
            st.set_option(k, v)
        st.upload(x)
        st.apply_op(op)
        return st.download()


def oracle_apply(O, n, op, x):
    out = np.zeros_like(x)
    O.apply_op_overwrite(n, op, x, out)
    return out


def check(O, n, op, seed=0, exact=False, bitwise=True, dtype=np.complex128, paths=("fast", "generic")):
    x = rand_state(n, seed, dtype)
    want = oracle_apply(O, n, op, x)
    tol = TOL64 if dtype == np.complex128 else TOL32
    for path in paths:
        opts = {"force_generic": 1} if path == "generic" else {}
        got = hip_apply(n, op, x, **opts)
        if exact or bitwise:
            assert np.array_equal(got, want), f"{op!r} n={n} path={path}: not IEEE-equal, max|d|={np.max(np.abs(got - want))}"
        else:
            assert np.max(np.abs(got - want)) <= tol, f"{op!r} n={n} path={path}"


def ref_twice(n, ops, x):
    with q.HipState(n) as st:
        st.upload(x)
        st.apply_ops(ops)
        st.apply_ops(ops)
        return st.download()


def _permuted(n, pi, x):
    j = np.arange(1 << n, dtype=np.uint64)
    src = np.zeros_like(j)
    for dbit in range(n):
        src |= ((j >> np.uint64(dbit)) & np.uint64(1)) << np.uint64(pi[dbit])
    return x[src.astype(np.int64)]


def _special_gates(n, rng):
    """one gate per kernel class / addressing corner, on the bit positions where launch shapes change"""
    u2 = rand_unitary(2, rng)
    u3 = rand_unitary(3, rng)
    u5 = rand_unitary(5, rng)
    ph = cmath.rect(1.0, 0.37)
    return [
        ("T_bit0", q.make_matrix_op([n - 1], circuits.T), True),
        ("H_bit0", q.make_matrix_op([n - 1], circuits.H), True),
        ("H_top", q.make_matrix_op([0], circuits.H), True),
        ("Rz_top", q.make_matrix_op([0], circuits.rz(0.77)), True),
        ("Rz_bit2", q.make_matrix_op([n - 3], circuits.rz(1.3)), True),
        ("cnot_lowctl", q.make_control_op([n - 1], q.make_matrix_op([0], circuits.X)), True),
        ("cnot_lowtgt", q.make_control_op([0], q.make_matrix_op([n - 2], circuits.X)), True),
        ("toffoli", q.make_control_op([0, n - 4], q.make_matrix_op([n // 2], circuits.X)), True),
        ("cphase", q.make_control_op([1], q.make_matrix_op([n - 1], [1, 0, 0, ph])), True),
        ("cH", q.make_control_op([n // 2], q.make_matrix_op([0], circuits.H)), True),
        ("swap1", q.make_swap_op([0], [n - 1]), True),
        ("swap2", q.make_swap_op([1, n - 8], [n - 2, 2]), True),
        ("dense2", q.make_matrix_op([0, n - 1], u2.ravel()), True),
        ("dense3_low_mfma", q.make_matrix_op([n - 1, n - 2, n - 3], u3.ravel()), False),
        ("dense3_high", q.make_matrix_op([0, 5, n - 9], u3.ravel()), True),
        ("dense5_mfma", q.make_matrix_op([0, 2, n - 20, n - 7, n - 1], u5.ravel()), False),
        ("cdense2", q.make_control_op([3], q.make_matrix_op([1, n - 5], u2.ravel())), True),
        ("dense7_streamed_mfma", q.make_matrix_op([0, 2, n - 20, n - 7, n - 1, 7, n - 12], rand_unitary(7, rng).ravel()), False),
        ("diag2", q.make_matrix_op([0, n - 2], np.diag([ph, ph.conjugate(), 1j, -1]).ravel()), True),
        ("sparse2", q.make_sparse_matrix_op([n - 1, 0], [[(0, 0.6), (1, 0.8j)], [(1, 0.6), (0, 0.8j)], [(3, 1j)], [(2, -1)]]), True),
        # k >= 6 SparseMatrix: the group staged in LDS beside the wave row (k_sparse_tile, r4) — all positions high / one in the row
        ("sparse6_two_per_row_tile", q.make_sparse_matrix_op([0, n // 2, 5, 7, 9, 12], [[(r, 0.6), (r ^ 9, 0.8j)] for r in range(64)]), True),
        ("sparse8_perm_phase_tile", q.make_sparse_matrix_op([0, 3, n // 2, 7, n - 1, 11, n - 9, 20],
                                                            [[(int(c), complex(np.exp(0.1j * r)))] for r, c in enumerate(np.random.default_rng(1).permutation(256))]), True),
        # (three stored entries per row, one of them a stored zero: nothing is filtered, and the op stays unitary)
        ("csparse6_tile", q.make_control_op([1, n - 2], q.make_sparse_matrix_op([0, n // 2, 5, n - 12, 9, n - 4], [[(r ^ 33, 0.8j), ((r * 7 + 3) % 64, 0.0), (r, 0.6)] for r in range(64)])), True),
    ]


def _run_dist(nproc, extra, worker="dist_worker_gpu.py", timeout=900):
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", worker)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return res.stdout


def _jit_info():
    import ctypes as C

    from rustqip_amd import _ffi

    k = C.c_uint64(_ffi.jit_counters()["kernels_resident_total"])
    res, ev, cap = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert _ffi.lib.qip_hip_jit_cache_info(C.byref(res), C.byref(ev), C.byref(cap)) == 0
    return {"compiled": int(k.value), "resident": int(res.value), "evicted": int(ev.value), "cap": int(cap.value)}


def _ansatz(n, thetas):
    """two layers of Rz / real rotations / controlled phases with one angle per qubit and layer, CNOT ladders between"""
    ops = []
    for layer in range(len(thetas)):
        for t in range(n):
            th = float(thetas[layer][t])
            ops.append(q.make_matrix_op([t], circuits.rz(th)))
            c, s = math.cos(th / 2), math.sin(th / 2)
            ops.append(q.make_matrix_op([(t + 3) % n], [c, -s, s, c]))
        for t in range(0, n - 1, 2):
            ops.append(q.make_control_op([t], q.make_matrix_op([t + 1], circuits.X)))
        for t in range(0, n - 2, 3):
            ops.append(q.make_control_op([t], q.make_matrix_op([t + 2], [1, 0, 0, cmath.rect(1, float(thetas[layer][t]) * 0.5)])))
    return ops


