"""Host-side mirror of the reference's gate descriptor and its validated constructors.

Same names, argument meaning and error behaviour as
  qip-iterators/src/iterators/ops.rs:11-91   (enum MatrixOp<P>, new_matrix/new_sparse/new_swap/new_control)
  qip/src/state_ops/matrix_ops.rs:12-122     (make_matrix_op, make_sparse_matrix_op, make_swap_op, make_control_op)
  qip/src/errors.rs:6-22                     (CircuitError)
so parity tests read like the reference's own.  Pure host bookkeeping: nothing here touches
amplitudes; descriptors are marshalled into `struct qip_op` for the C ABI by `to_c`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from enum import Enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi


class CircuitError(Exception):
    """qip::errors::CircuitError::Generic(String) (qip/src/errors.rs:6-22)."""


class Representation(Enum):
    """qip/src/types.rs:16-22"""

    BigEndian = 0
    LittleEndian = 1


_NP_OF_DTYPE = {_ffi.QIP_C64: np.complex128, _ffi.QIP_C32: np.complex64, _ffi.QIP_F64: np.float64, _ffi.QIP_F32: np.float32,
                _ffi.QIP_I64: np.int64, _ffi.QIP_I32: np.int32}


def complex_dtype(dtype: int):
    return np.complex128 if dtype == _ffi.QIP_C64 else np.complex64


def element_dtype(dtype: int):
    """numpy type of one element `P` of enum qip_dtype (complex for a state; real / integer for the slice-level calls)"""
    return _NP_OF_DTYPE[dtype]


def _payload(values, dtype: int) -> np.ndarray:
    """op payload (Matrix data / SparseMatrix values) as `P`: a real or integer `P` takes no imaginary part and no fraction"""
    et = element_dtype(dtype)
    a = np.asarray(values)
    if not np.issubdtype(et, np.complexfloating):
        if np.iscomplexobj(a):
            if np.any(a.imag != 0):
                raise CircuitError("op payload has an imaginary part but the vectors are real")
            a = a.real
        if np.issubdtype(et, np.integer) and np.any(a != np.floor(a)):
            raise CircuitError("op payload has a fractional part but the vectors are integers")
    return np.ascontiguousarray(a, dtype=et)


def flip_bits(n: int, num: int) -> int:
    """qip-iterators/src/utils.rs:22-25: reverse the low n bits of num."""
    out = 0
    for b in range(n):
        if (num >> b) & 1:
            out |= 1 << (n - 1 - b)
    return out


@dataclass
class MatrixOp:
    """MatrixOp<P> (ops.rs:11-20).  kind is one of "Matrix", "SparseMatrix", "Swap", "Control"."""

    kind: str
    indices: List[int]
    data: Optional[np.ndarray] = None  # Matrix: 4^k complex entries, row-major
    rows: Optional[List[List[Tuple[int, complex]]]] = None  # SparseMatrix: per row [(col, val)]
    half: int = 0  # Swap: number of A (= B) indices
    n_controls: int = 0  # Control
    inner: Optional["MatrixOp"] = None  # Control
    _keep: list = field(default_factory=list, repr=False, compare=False)

    # ---- ops.rs:49-91 ---------------------------------------------------------
    @staticmethod
    def new_matrix(indices: Sequence[int], data) -> "MatrixOp":
        return MatrixOp("Matrix", list(indices), data=np.asarray(data, dtype=np.complex128).ravel())

    @staticmethod
    def new_sparse(indices: Sequence[int], rows) -> "MatrixOp":
        return MatrixOp("SparseMatrix", list(indices), rows=[[(int(c), complex(v)) for c, v in r] for r in rows])

    @staticmethod
    def new_swap(a: Sequence[int], b: Sequence[int]) -> "MatrixOp":
        a, b = list(a), list(b)
        return MatrixOp("Swap", a + b, half=len(a))

    @staticmethod
    def new_control(c: Sequence[int], r: Sequence[int], op: "MatrixOp") -> "MatrixOp":
        c = list(c)
        return MatrixOp("Control", c + list(r), n_controls=len(c), inner=op)

    # ---- ops.rs:24-46 ---------------------------------------------------------
    def num_indices(self) -> int:
        return 2 * self.half if self.kind == "Swap" else len(self.indices)

    def get_index(self, i: int) -> int:
        """qip_iterators::matrix_ops::get_index (matrix_ops.rs:33-35)"""
        return self.indices[i]

    # ---- marshalling ------------------------------------------------------------
    def to_c(self, dtype: int = _ffi.QIP_C64) -> _ffi.QipOp:
        """Build the `struct qip_op` tree; the returned object keeps every buffer alive."""
        op = _ffi.QipOp()
        keep: list = []
        idx = np.ascontiguousarray(self.indices, dtype=np.uint64)
        keep.append(idx)
        op.n_indices = len(self.indices)
        op.indices = idx.ctypes.data_as(C.POINTER(C.c_uint64))
        if self.kind == "Matrix":
            op.kind = _ffi.QIP_OP_MATRIX
            k = len(self.indices)
            if self.data is None or self.data.size != 4**k:
                raise CircuitError(
                    f"Matrix data has {0 if self.data is None else self.data.size} entries versus expected 2^2*{k}"
                )
            dat = _payload(self.data, dtype)
            keep.append(dat)
            op.dense = dat.ctypes.data
        elif self.kind == "SparseMatrix":
            op.kind = _ffi.QIP_OP_SPARSE
            k = len(self.indices)
            if self.rows is None or len(self.rows) != 1 << k:
                raise CircuitError(
                    f"Sparse matrix has {0 if self.rows is None else len(self.rows)} rows versus expected 2^{k}"
                )
            rowptr = np.zeros(len(self.rows) + 1, dtype=np.uint64)
            cols: list = []
            vals: list = []
            for r, row in enumerate(self.rows):
                for c, v in row:
                    cols.append(c)
                    vals.append(v)
                rowptr[r + 1] = len(cols)
            cols_a = np.ascontiguousarray(cols, dtype=np.uint64)
            vals_a = _payload(vals, dtype)
            keep += [rowptr, cols_a, vals_a]
            op.sparse_rowptr = rowptr.ctypes.data_as(C.POINTER(C.c_uint64))
            op.sparse_cols = cols_a.ctypes.data_as(C.POINTER(C.c_uint64))
            op.sparse_vals = vals_a.ctypes.data
        elif self.kind == "Swap":
            op.kind = _ffi.QIP_OP_SWAP
        elif self.kind == "Control":
            op.kind = _ffi.QIP_OP_CONTROL
            op.n_controls = self.n_controls
            if self.inner is None:
                raise CircuitError("Control op without inner op")
            inner_c = self.inner.to_c(dtype)
            keep.append(inner_c)
            op.inner = C.pointer(inner_c)
        else:
            raise CircuitError(f"unknown op kind {self.kind!r}")
        op._keep = keep  # ctypes Structures accept ad-hoc attributes; ties buffer lifetime to the struct
        return op

    def __repr__(self) -> str:  # ops.rs:159-181 Debug
        if self.kind == "Control":
            return f"C({self.inner!r})[{', '.join(map(str, self.indices[: self.n_controls]))}]"
        return f"{self.kind}[{', '.join(map(str, self.indices))}]"


# ---- qip/src/state_ops/matrix_ops.rs constructors --------------------------------


def make_matrix_op(indices: Sequence[int], dat) -> MatrixOp:
    """make_matrix_op (matrix_ops.rs:12-27)"""
    indices = list(indices)
    n = len(indices)
    dat = np.asarray(dat, dtype=np.complex128).ravel()
    if n == 0:
        raise CircuitError("Must supply at least one op index")
    if dat.size != 1 << (2 * n):
        raise CircuitError(f"Matrix data has {dat.size} entries versus expected 2^2*{n}")
    return MatrixOp("Matrix", indices, data=dat)


def make_sparse_matrix_op(indices: Sequence[int], dat, order: Representation = Representation.BigEndian) -> MatrixOp:
    """make_sparse_matrix_op (matrix_ops.rs:32-81), including the little-endian -> big-endian flip (:62-77)."""
    indices = list(indices)
    n = len(indices)
    rows = [[(int(c), complex(v)) for c, v in r] for r in dat]
    if n == 0:
        raise CircuitError("Must supply at least one op index")
    if len(rows) != 1 << n:
        raise CircuitError(f"Sparse matrix has {len(rows)} rows versus expected 2^{n}")
    for r, v in enumerate(rows):
        if not v:
            raise CircuitError(f"All rows of sparse matrix must have data ({r} is empty)")
    if order is Representation.LittleEndian:
        flipped = [(i, [(flip_bits(n, c), v) for c, v in r]) for i, r in enumerate(rows)]
        flipped.sort(key=lambda t: flip_bits(n, t[0]))  # stable, like sort_by_key
        rows = [r for _, r in flipped]
    return MatrixOp("SparseMatrix", indices, rows=rows)


def make_swap_op(a_indices: Sequence[int], b_indices: Sequence[int]) -> MatrixOp:
    """make_swap_op (matrix_ops.rs:84-100)"""
    a, b = list(a_indices), list(b_indices)
    if not a or not b:
        raise CircuitError("Need at least 1 swap index for a and b")
    if len(a) != len(b):
        raise CircuitError(
            f"Swap must be performed on two sets of indices of equal length, found {len(a)} vs {len(b)}"
        )
    return MatrixOp("Swap", a + b, half=len(a))


def make_control_op(c_indices: Sequence[int], op: MatrixOp) -> MatrixOp:
    """make_control_op (matrix_ops.rs:103-122): nested controls are collapsed (:112-115)."""
    c = list(c_indices)
    if not c:
        raise CircuitError("Must supply at least one control index")
    if op.kind == "Control":
        return MatrixOp("Control", c + list(op.indices), n_controls=len(c) + op.n_controls, inner=op.inner)
    return MatrixOp("Control", c + list(op.indices), n_controls=len(c), inner=op)


def flatten(op: MatrixOp):
    """(control qubits, innermost op, target qubits) with nested Controls accumulated the way
    sum_for_control_iterator does (ops.rs:150-154); only the OUTER index list is used (matrix_ops.rs:108)."""
    if op.kind != "Control":
        return [], op, list(op.indices)
    n_control, inner = op.n_controls, op.inner
    n_op = len(op.indices) - op.n_controls
    while inner.kind == "Control":
        n_control += inner.n_controls
        n_op = len(inner.indices) - inner.n_controls
        inner = inner.inner
    if n_control + n_op != len(op.indices):
        raise CircuitError("Control op index list does not match its controls + inner op indices")
    return list(op.indices[:n_control]), inner, list(op.indices[n_control:])


def validate_op(n: int, op: MatrixOp, dtype: int = _ffi.QIP_C64) -> None:
    """Run the C ABI validator (host code, no GPU needed); raises CircuitError."""
    cop = op.to_c(dtype)
    rc = _ffi.lib.qip_hip_validate_op(n, C.byref(cop))
    if rc != _ffi.QIP_OK:
        raise CircuitError(_ffi.last_error())


def algorithmic_bytes(n: int, op: MatrixOp, dtype: int = _ffi.QIP_C64) -> float:
    """Algorithmic bytes of one application; host code, no GPU needed."""
    cop = op.to_c(dtype)
    out = C.c_double(0)
    rc = _ffi.lib.qip_hip_op_algorithmic_bytes(dtype, n, C.byref(cop), C.byref(out))
    if rc != _ffi.QIP_OK:
        raise CircuitError(_ffi.last_error())
    return out.value


def plan_tiles(n: int, ops, mode: int = 1, dtype: int = _ffi.QIP_C64):
    """Schedule of the LDS-resident multi-gate sweeps (option "tile") for a circuit — host code, no GPU.
    Returns a list of steps, each a list of op indices in application order."""
    cops = [op.to_c(dtype) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    step_of = (C.c_int64 * len(cops))()
    n_steps = C.c_uint64()
    rc = _ffi.lib.qip_hip_plan_tiles(dtype, n, arr, len(cops), mode, step_of, C.byref(n_steps))
    if rc != _ffi.QIP_OK:
        raise CircuitError(_ffi.last_error())
    steps = [[] for _ in range(n_steps.value)]
    for i in range(len(cops)):
        if step_of[i] >= 0:  # mode bit 2 (relabelling): an uncontrolled Swap op became a label exchange, it is in no step
            steps[step_of[i]].append(i)
    return steps


TILE_BITS = int(_ffi.lib.qip_hip_tile_bits())  # kTileBits of csrc/qip_kernels.h: low 6 index bits + the free positions
TILE_LANE_BITS = TILE_BITS - 3                 # thread-id bits of a tile block (a lane holds 2^3 elements)


def tile_lane_assignment(pass_bits, dtype: int = _ffi.QIP_C64):
    """Host-only: for one pass of a tile sweep (its three exchange bits, tile-index space, ascending), the
    tile-index bit each of the 8 thread-id bits fills (qip_hip_tile_lane_assignment)."""
    pb = (C.c_uint32 * 3)(*pass_bits)
    out = C.c_uint64()
    rc = _ffi.lib.qip_hip_tile_lane_assignment(dtype, pb, C.byref(out))
    if rc != _ffi.QIP_OK:
        raise CircuitError(_ffi.last_error())
    return [(out.value >> (4 * k)) & 15 for k in range(TILE_LANE_BITS)]


def debug_tile_jit(n: int, ops, mode: int = 1, dtype: int = _ffi.QIP_C64) -> dict:
    """Host-only test hook (qip_hip_debug_tile_jit): write out and hiprtc-compile the run-time-specialised kernel of
    every multi-gate tile segment of the circuit (no GPU needed)."""
    cops = [op.to_c(dtype) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    nseg, nsrc, ncode = C.c_uint64(), C.c_uint64(), C.c_uint64()
    first = C.c_char_p()
    rc = _ffi.lib.qip_hip_debug_tile_jit(dtype, n, arr, len(cops), mode, C.byref(nseg), C.byref(nsrc), C.byref(ncode), C.byref(first))
    if rc != _ffi.QIP_OK:
        raise CircuitError(_ffi.last_error())
    return {"segments": int(nseg.value), "source_bytes": int(nsrc.value), "code_bytes": int(ncode.value),
            "first_source": (first.value or b"").decode()}


def debug_sparse_tile(n: int, op, dtype: int = _ffi.QIP_C64) -> dict:
    """Host-only test hook (qip_hip_debug_sparse_tile): what the host ships to k_sparse_tile for one SparseMatrix op, or
    {"applies": 0} when the op takes another kernel."""
    import json

    cop = op.to_c(dtype)
    arr = (_ffi.QipOp * 1)(cop)
    txt = _ffi.lib.qip_hip_debug_sparse_tile(dtype, n, arr)
    if not txt:
        raise CircuitError(_ffi.last_error())
    return json.loads(txt.decode() if isinstance(txt, bytes) else txt)


def debug_tile_plan(n: int, ops, mode: int = 1, dtype: int = _ffi.QIP_C64) -> dict:
    """Host-only test hook (qip_hip_debug_tile_plan): the tile schedule plus every segment's passes and gate
    descriptors as shipped to the kernel, parsed from JSON."""
    import json

    cops = [op.to_c(dtype) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    txt = _ffi.lib.qip_hip_debug_tile_plan(dtype, n, arr, len(cops), mode)
    if not txt:
        raise CircuitError(_ffi.last_error())
    return json.loads(txt.decode() if isinstance(txt, bytes) else txt)
