"""TEST INFRASTRUCTURE — oracle comparison at sizes where the full vector cannot be compared.

SURVEY.md §8(d): "at full n, … a window compare where the oracle can be evaluated (apply_op_row on
selected rows is cheap: matrix_ops.rs:38-59)".  A row of `op · x` only reads the columns that differ from the
row on the op's own bit positions (matrix_ops.rs:74-93, sub_to_full :24-30).  So for a window of 2^w
consecutive rows the needed inputs are the <= 2^h windows obtained by setting the op's h bit positions >= w
every possible way: together they form a sub-cube of the index space that is CLOSED under the op.  On that
sub-cube the op acts exactly like the same MatrixOp on an m = w + h qubit register (same matrix rows, same
column order, hence the same sequence of rounded operations per row), which is what the oracle evaluates:

    before  = the 2^h windows downloaded from the device before the gate         (2^m amplitudes)
    want    = qip_oracle apply_op_overwrite(m, op', before)   and, on sampled rows, qip_oracle apply_op_row
    after   = the same windows downloaded after the gate

Only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this module; the product never does.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from rustqip_amd.ops import MatrixOp


def _remap(op: MatrixOp, qmap: Dict[int, int]) -> MatrixOp:
    inner = _remap(op.inner, qmap) if op.inner is not None else None
    return MatrixOp(op.kind, [qmap.get(int(q), 0) for q in op.indices], data=op.data, rows=op.rows, half=op.half,
                    n_controls=op.n_controls, inner=inner)


class SubCube:
    """Index bits `low` = 0..w-1 plus the sorted positions `high`; every other bit is fixed by `base`."""

    def __init__(self, n: int, positions: Sequence[int], w_max: int = 16, m_max: int = 22):
        pos = sorted(set(int(p) for p in positions))
        w = min(w_max, n)
        while w > 0 and w + sum(1 for p in pos if p >= w) > min(m_max, n):
            w -= 1
        self.n, self.w = n, w
        self.high = [p for p in pos if p >= w]
        self.m = w + len(self.high)
        self.ok = w >= 4 and self.m <= m_max
        self.vmask = ((1 << w) - 1) | sum(1 << p for p in self.high)

    def sub_position(self, p: int) -> int:
        return p if p < self.w else self.w + self.high.index(p)

    def offsets(self, base: int) -> List[int]:
        base &= ~self.vmask & ((1 << self.n) - 1)
        out = []
        for h in range(1 << len(self.high)):
            off = base
            for j, p in enumerate(self.high):
                off |= ((h >> j) & 1) << p
            out.append(off)
        return out

    def gather(self, download: Callable[[int, int], np.ndarray], base: int) -> np.ndarray:
        return np.concatenate([download(off, 1 << self.w) for off in self.offsets(base)])

    def localize(self, op: MatrixOp) -> MatrixOp:
        qmap = {}
        for q in op.indices:
            qmap[int(q)] = self.m - 1 - self.sub_position(self.n - 1 - int(q))
        return _remap(op, qmap)


def default_bases(n: int, seed: int = 0, count: int = 4) -> List[int]:
    """bottom and top of the index space plus seeded random places in between"""
    rng = np.random.default_rng(seed)
    full = (1 << n) - 1
    bases = [0, full]
    while len(bases) < count:
        bases.append(int(rng.integers(0, 1 << 62)) & full)
    return bases[:count]


def check_ops(state, n: int, ops: Sequence[MatrixOp], O, bases: Optional[Sequence[int]] = None,
              apply: Optional[Callable[[], None]] = None, rows_sampled: int = 8, w_max: int = 16) -> Optional[dict]:
    """Apply `ops` to the device state (state.apply_ops, or `apply()`), and compare >= len(bases) sub-cubes of
    2^m amplitudes against the oracle applied to what the device held before.  Returns None when the ops touch
    too many high bit positions for a sub-cube of reasonable size (nothing is applied then), else
    {"rows": amplitudes compared, "max_abs_delta": ..., "bit_equal": every compared component IEEE-==,
     "row_calls": rows additionally recomputed with apply_op_row, "m": ..., "windows": ...}."""
    ops = list(ops)
    positions = [n - 1 - int(q) for op in ops for q in op.indices]
    cube = SubCube(n, positions, w_max=w_max)
    if not cube.ok:
        return None
    bases = list(bases) if bases is not None else default_bases(n)
    before = [cube.gather(state.download, b) for b in bases]
    if apply is not None:
        apply()
    else:
        state.apply_ops(ops)
    local_ops = [cube.localize(op) for op in ops]
    worst, equal, rows, row_calls = 0.0, True, 0, 0
    rng = np.random.default_rng(n)
    for b, x in zip(bases, before):
        got = cube.gather(state.download, b)
        cur, arena = x.copy(), np.zeros_like(x)
        for lop in local_ops:
            O.apply_op_overwrite(cube.m, lop, cur, arena)
            cur, arena = arena, cur
        d = np.abs(got - cur)
        worst = max(worst, float(d.max()))
        equal = equal and bool(np.array_equal(got, cur))
        rows += got.size
        if len(local_ops) == 1:
            # the literal per-row entry point of the reference (matrix_ops.rs:38-59) on sampled rows, first and last included
            sample = [0, got.size - 1] + [int(r) for r in rng.integers(0, got.size, size=rows_sampled)]
            for r in sample:
                v = O.apply_op_row(cube.m, local_ops[0], x, r)
                worst = max(worst, abs(complex(got[r]) - v))
                equal = equal and (complex(got[r]) == v)
                row_calls += 1
    return {"rows": rows, "max_abs_delta": worst, "bit_equal": equal, "row_calls": row_calls, "m": cube.m,
            "windows": len(bases) * (1 << len(cube.high))}


def chunk_by_high_bits(n: int, ops: Sequence[MatrixOp], max_high: int = 6, w: int = 16, max_len: int = 64) -> List[List[MatrixOp]]:
    """Cut a circuit into consecutive chunks whose ops together touch at most `max_high` bit positions >= w, so
    each chunk has a closed sub-cube of <= 2^(w + max_high) amplitudes."""
    chunks: List[List[MatrixOp]] = []
    cur: List[MatrixOp] = []
    high: set = set()
    for op in ops:
        mine = {n - 1 - int(q) for q in op.indices if n - 1 - int(q) >= w}
        if cur and (len(high | mine) > max_high or len(cur) >= max_len):
            chunks.append(cur)
            cur, high = [], set()
        cur.append(op)
        high |= mine
    if cur:
        chunks.append(cur)
    return chunks


def check_circuit(state, n: int, ops: Sequence[MatrixOp], O, gate_by_gate: bool = True, seed: int = 0,
                  bases_per_step: int = 4) -> dict:
    """check_ops over a whole circuit: gate by gate, or in chunks applied through state.apply_ops (which is how the
    multi-gate tile sweeps are reached).  Aggregates the per-step results."""
    steps = [[op] for op in ops] if gate_by_gate else chunk_by_high_bits(n, ops)
    agg = {"gates": 0, "steps": 0, "rows": 0, "row_calls": 0, "max_abs_delta": 0.0, "bit_equal": True, "skipped": 0,
           "windows": 0}
    for i, chunk in enumerate(steps):
        r = check_ops(state, n, chunk, O, bases=default_bases(n, seed + i, bases_per_step))
        if r is None:
            state.apply_ops(chunk)
            agg["skipped"] += len(chunk)
            continue
        agg["gates"] += len(chunk)
        agg["steps"] += 1
        agg["rows"] += r["rows"]
        agg["row_calls"] += r["row_calls"]
        agg["windows"] += r["windows"]
        agg["max_abs_delta"] = max(agg["max_abs_delta"], r["max_abs_delta"])
        agg["bit_equal"] = agg["bit_equal"] and r["bit_equal"]
    return agg


def product_state_ops(n: int, seed: int):
    """Ops that turn |0..0> into a seeded PRODUCT state whose amplitudes are pairwise distinct in modulus and phase
    (H, Rz(theta_t), Ry-like real rotation, Rz(phi_t) per qubit), and the per-qubit 2-vectors for its closed form
    amp(idx) = prod_t v[t][bit_t(idx)].  A uniform state hides index-mapping errors; this one does not."""
    import cmath
    import math

    from rustqip_amd.ops import make_matrix_op

    rng = np.random.default_rng(seed)
    ops, vecs = [], []
    for t in range(n):
        a = float(rng.uniform(0.55, 1.0))  # rotation angle: both components well away from 0
        ph0, ph1 = float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(0, 2 * math.pi))
        m = np.array([[math.cos(a) * cmath.rect(1, ph0), -math.sin(a)], [math.sin(a) * cmath.rect(1, ph1), math.cos(a)]],
                     dtype=np.complex128)
        ops.append(make_matrix_op([t], m.ravel()))
        vecs.append((complex(m[0, 0]), complex(m[1, 0])))
    return ops, vecs


def product_state_window(n: int, vecs, offset: int, length: int) -> np.ndarray:
    """closed form of the product state on [offset, offset+length) (qubit t <-> index bit n-1-t)"""
    idx = np.arange(offset, offset + length, dtype=np.uint64)
    out = np.ones(length, dtype=np.complex128)
    for t, (v0, v1) in enumerate(vecs):
        bit = (idx >> np.uint64(n - 1 - t)) & np.uint64(1)
        out *= np.where(bit == 1, v1, v0)
    return out
