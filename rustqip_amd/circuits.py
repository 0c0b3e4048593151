"""Synthetic circuits at MatrixOp level: the workloads of BASELINE.json's configs
(definitions in SURVEY.md §8(d)).  Pure host code; the same generators feed the parity tests
(at reduced n) and bench.py (at full n).

Gate matrices are the ones the reference run loop lowers to (qip/src/builder.rs:436-498).
"""
from __future__ import annotations

import cmath
import math
from typing import List

import numpy as np

from .ops import MatrixOp, make_control_op, make_matrix_op, make_swap_op

_S = math.sqrt(0.5)
X = [0, 1, 1, 0]
Z = [1, 0, 0, -1]
H = [complex(_S, 0.0), complex(_S, 0.0), complex(_S, 0.0), -complex(_S, 0.0)]
S = [1, 0, 0, 1j]
T = [1, 0, 0, cmath.rect(1.0, math.pi / 4)]


def rz(theta: float):
    return [cmath.rect(1.0, -theta / 2), 0, 0, cmath.rect(1.0, theta / 2)]


def h_layer(n: int) -> List[MatrixOp]:
    return [make_matrix_op([q], H) for q in range(n)]


def random_state(n: int, seed: int, dtype=np.complex128) -> np.ndarray:
    """Seeded random normalised state: re, im ~ N(0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = 1 << n
    v = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    v /= np.linalg.norm(v)
    return v.astype(dtype)


def c2_random_circuit(n: int, n_gates: int = 256, seed: int = 28, single_only: bool = False) -> List[MatrixOp]:
    """configs[1]: random single-qubit (H / X / Rz) + CNOT circuit.  Each gate is single-qubit
    with probability 3/4 (uniform over {H, X, Rz(theta~U[0,2pi))}, uniform target), else a CNOT
    on a uniform ordered pair c != t."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ops: List[MatrixOp] = []
    for _ in range(n_gates):
        if single_only or rng.random() < 0.75:
            kind = int(rng.integers(0, 3))
            q = int(rng.integers(0, n))
            if kind == 0:
                ops.append(make_matrix_op([q], H))
            elif kind == 1:
                ops.append(make_matrix_op([q], X))
            else:
                ops.append(make_matrix_op([q], rz(float(rng.random()) * 2 * math.pi)))
        else:
            c = int(rng.integers(0, n))
            t = int(rng.integers(0, n - 1))
            t = t if t < c else t + 1
            ops.append(make_control_op([c], make_matrix_op([t], X)))
    return ops


def c3_qft(n: int) -> List[MatrixOp]:
    """configs[2]: textbook QFT — for i: H(i); for j>i: controlled-phase(pi/2^(j-i)) with control
    j on target i; then n/2 swaps (i, n-1-i).  Built from native Control ops, not qfft()
    (SURVEY.md §0.3 / Appendix C Q2)."""
    ops: List[MatrixOp] = []
    for i in range(n):
        ops.append(make_matrix_op([i], H))
        for j in range(i + 1, n):
            ph = cmath.rect(1.0, math.pi / (1 << (j - i)))
            ops.append(make_control_op([j], make_matrix_op([i], [1, 0, 0, ph])))
    for i in range(n // 2):
        ops.append(make_swap_op([i], [n - 1 - i]))
    return ops


def c4_clifford_t(n: int, n_gates: int = 256, seed: int = 32) -> List[MatrixOp]:
    """configs[3]: random Clifford+T circuit, gates uniform over {H, S, T, CNOT}, uniform targets."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ops: List[MatrixOp] = []
    for _ in range(n_gates):
        kind = int(rng.integers(0, 4))
        q = int(rng.integers(0, n))
        if kind == 0:
            ops.append(make_matrix_op([q], H))
        elif kind == 1:
            ops.append(make_matrix_op([q], S))
        elif kind == 2:
            ops.append(make_matrix_op([q], T))
        else:
            t = int(rng.integers(0, n - 1))
            t = t if t < q else t + 1
            ops.append(make_control_op([q], make_matrix_op([t], X)))
    return ops


def c5_grover_iteration(n: int, dense_k3: bool = False) -> List[MatrixOp]:
    """configs[4]: one Grover iteration for the marked item |0...0>:
    mark = X^n · C^{n-1}Z · X^n (phase flip of the marked item); diffusion = H^n · X^n · C^{n-1}Z · X^n · H^n (6n+2 ops).
    With dense_k3 the H (and X) on the three lowest-bit-position qubits n-3..n-1 are merged into
    one 8x8 Matrix op (the dense k=3 path)."""
    def layer(m):
        if dense_k3 and n >= 3:
            m2 = np.asarray(m, dtype=np.complex128).reshape(2, 2)
            m8 = np.kron(np.kron(m2, m2), m2)
            return [make_matrix_op([q], m) for q in range(n - 3)] + [make_matrix_op([n - 3, n - 2, n - 1], m8.ravel())]
        return [make_matrix_op([q], m) for q in range(n)]

    def mcz():
        if n == 1:
            return [make_matrix_op([0], Z)]
        return [make_control_op(list(range(n - 1)), make_matrix_op([n - 1], Z))]

    ops: List[MatrixOp] = []
    ops += layer(X) + mcz() + layer(X)
    ops += layer(H) + layer(X) + mcz() + layer(X) + layer(H)
    return ops
