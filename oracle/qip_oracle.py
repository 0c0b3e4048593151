"""TEST INFRASTRUCTURE — Python face of the CPU oracle (oracle/qip_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It provides
  * ctypes bindings of the C restatement of apply_op / apply_op_overwrite / measurement
    (each C function cites the reference file:line it follows), and
  * an independent restatement of the reference run loop: initial basis index, gate -> MatrixOp
    lowering table, ping-pong application, measurement dispatch
    (qip/src/builder.rs:400-519).
Descriptors are passed as the same `struct qip_op` the product's C ABI takes, so the HIP path
and the oracle see bit-identical inputs.
"""
from __future__ import annotations

import cmath
import ctypes as C
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqip_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("qip_oracle.c", "qip_oracle_impl.h", "qip_oracle_real_impl.h")]
    src.append(os.path.join(_HERE, "..", "include", "qip_hip.h"))
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B", "libqip_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return _LIB


build()
_lib = C.CDLL(_LIB)

# the op struct/marshalling is shared with the product host code (descriptor only, no compute)
from rustqip_amd import _ffi as _pffi  # noqa: E402
from rustqip_amd.ops import MatrixOp  # noqa: E402

_opp = C.POINTER(_pffi.QipOp)
_u64p = C.POINTER(C.c_uint64)


class _C64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class _C32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


for suf, cplx, real in (("c64", _C64, C.c_double), ("c32", _C32, C.c_float)):
    f = getattr(_lib, f"qip_oracle_apply_op_{suf}")
    f.restype = None
    f.argtypes = [C.c_uint32, _opp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                  C.c_int, C.c_int]
    f = getattr(_lib, f"qip_oracle_apply_op_row_{suf}")
    f.restype = cplx
    f.argtypes = [C.c_uint32, _opp, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    f = getattr(_lib, f"qip_oracle_prob_magnitude_{suf}")
    f.restype = real
    f.argtypes = [C.c_void_p, C.c_uint64]
    f = getattr(_lib, f"qip_oracle_measure_prob_{suf}")
    f.restype = real
    f.argtypes = [C.c_uint32, C.c_uint64, _u64p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64]
    f = getattr(_lib, f"qip_oracle_measure_probs_{suf}")
    f.restype = None
    f.argtypes = [C.c_uint32, _u64p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    f = getattr(_lib, f"qip_oracle_soft_measure_{suf}")
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint32, _u64p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double]
    f = getattr(_lib, f"qip_oracle_measure_state_{suf}")
    f.restype = C.c_int
    f.argtypes = [C.c_uint32, _u64p, C.c_uint32, C.c_uint64, real, C.c_void_p, C.c_uint64, C.c_void_p,
                  C.c_uint64, C.c_uint64, C.c_uint64]

# real / integer P (qip_oracle_real_impl.h): apply_op / apply_op_overwrite / apply_op_row only
for suf, elem in (("f64", C.c_double), ("f32", C.c_float), ("i64", C.c_int64), ("i32", C.c_int32)):
    f = getattr(_lib, f"qip_oracle_apply_op_{suf}")
    f.restype = None
    f.argtypes = [C.c_uint32, _opp, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                  C.c_int, C.c_int]
    f = getattr(_lib, f"qip_oracle_apply_op_row_{suf}")
    f.restype = elem
    f.argtypes = [C.c_uint32, _opp, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]

for name, res, args in (
    ("qip_oracle_get_flat_index", C.c_uint64, [C.c_uint32, C.c_uint64, C.c_uint64]),
    ("qip_oracle_flip_bits", C.c_uint64, [C.c_uint32, C.c_uint64]),
    ("qip_oracle_set_bit", C.c_uint64, [C.c_uint64, C.c_uint32, C.c_int]),
    ("qip_oracle_get_bit", C.c_int, [C.c_uint64, C.c_uint32]),
    ("qip_oracle_entwine_bits", C.c_uint64, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]),
    ("qip_oracle_extract_bits", C.c_uint64, [C.c_uint64, _u64p, C.c_uint32]),
    ("qip_oracle_full_to_sub", C.c_uint64, [C.c_uint32, _u64p, C.c_uint32, C.c_uint64]),
    ("qip_oracle_sub_to_full", C.c_uint64, [C.c_uint32, _u64p, C.c_uint32, C.c_uint64, C.c_uint64]),
    ("qip_oracle_max_threads", C.c_int, []),
):
    f = getattr(_lib, name)
    f.restype = res
    f.argtypes = args


def _u64(values: Sequence[int]):
    return (C.c_uint64 * len(values))(*[int(v) for v in values])


_ELEM = {np.dtype(np.complex128): ("c64", _pffi.QIP_C64), np.dtype(np.complex64): ("c32", _pffi.QIP_C32),
         np.dtype(np.float64): ("f64", _pffi.QIP_F64), np.dtype(np.float32): ("f32", _pffi.QIP_F32),
         np.dtype(np.int64): ("i64", _pffi.QIP_I64), np.dtype(np.int32): ("i32", _pffi.QIP_I32)}


def _suf(arr: np.ndarray) -> str:
    try:
        return _ELEM[arr.dtype][0]
    except KeyError:
        raise TypeError(f"oracle supports complex128/64, float64/32, int64/32, got {arr.dtype}") from None


def _dt(arr: np.ndarray) -> int:
    return _ELEM[arr.dtype][1]


# ---- bit utilities -------------------------------------------------------------------
def flip_bits(n, num): return int(_lib.qip_oracle_flip_bits(n, num))
def set_bit(num, bit_index, value): return int(_lib.qip_oracle_set_bit(num, bit_index, int(bool(value))))
def get_bit(num, bit_index): return bool(_lib.qip_oracle_get_bit(num, bit_index))
def get_flat_index(nindices, i, j): return int(_lib.qip_oracle_get_flat_index(nindices, i, j))
def entwine_bits(n, selector, off_bits, on_bits): return int(_lib.qip_oracle_entwine_bits(n, selector, off_bits, on_bits))
def extract_bits(num, indices): return int(_lib.qip_oracle_extract_bits(num, _u64(indices), len(indices)))
def full_to_sub(n, mat_indices, full_index): return int(_lib.qip_oracle_full_to_sub(n, _u64(mat_indices), len(mat_indices), full_index))
def sub_to_full(n, mat_indices, sub_index, base): return int(_lib.qip_oracle_sub_to_full(n, _u64(mat_indices), len(mat_indices), sub_index, base))
def max_threads(): return int(_lib.qip_oracle_max_threads())


def set_num_threads(n: int) -> None:
    """OpenMP threads of the following oracle calls (bench.py's CPU baseline times 1 thread and all threads)"""
    _lib.qip_oracle_set_num_threads(int(n))


# ---- the kernel ------------------------------------------------------------------------
def apply_op(n: int, op: MatrixOp, input: np.ndarray, output: np.ndarray, input_offset: int = 0,
             output_offset: int = 0, *, accumulate: bool = True, nthreads: int = 0) -> None:
    """qip_iterators::matrix_ops::apply_op (matrix_ops.rs:98-123); accumulate=False is
    apply_op_overwrite (:127-152)."""
    assert input.flags.c_contiguous and output.flags.c_contiguous and input.dtype == output.dtype
    cop = op.to_c(_dt(output))
    getattr(_lib, f"qip_oracle_apply_op_{_suf(output)}")(
        n, C.byref(cop), input.ctypes.data, input.size, output.ctypes.data, output.size, input_offset,
        output_offset, int(accumulate), nthreads)


def apply_op_overwrite(n, op, input, output, input_offset=0, output_offset=0, *, nthreads: int = 0) -> None:
    apply_op(n, op, input, output, input_offset, output_offset, accumulate=False, nthreads=nthreads)


def apply_op_row(n: int, op: MatrixOp, input: np.ndarray, outputrow: int, input_offset: int = 0,
                 output_offset: int = 0):
    """apply_op_row (matrix_ops.rs:38-59): a complex for a complex vector, a float / int for a real / integer one"""
    cop = op.to_c(_dt(input))
    r = getattr(_lib, f"qip_oracle_apply_op_row_{_suf(input)}")(
        n, C.byref(cop), input.ctypes.data, input.size, outputrow, input_offset, output_offset)
    return complex(r.re, r.im) if np.iscomplexobj(input) else r


def make_op_matrix(n: int, op: MatrixOp, dtype=np.complex128) -> np.ndarray:
    """The reference's own test method (matrix_ops.rs:229-255): column i = apply_op on basis i."""
    N = 1 << n
    cols = []
    for i in range(N):
        inp = np.zeros(N, dtype=dtype)
        out = np.zeros(N, dtype=dtype)
        inp[i] = 1
        apply_op(n, op, inp, out)
        cols.append(out)
    return np.stack(cols, axis=1)


def apply_ops_in_place(n: int, ops: Sequence[MatrixOp], state: np.ndarray, nthreads: int = 0) -> np.ndarray:
    """The unitary part of the run loop: apply_op_overwrite into the arena, swap (builder.rs:499,514)."""
    arena = np.zeros_like(state)
    for op in ops:
        apply_op_overwrite(n, op, state, arena, nthreads=nthreads)
        state, arena = arena, state
    return state


# ---- measurement -----------------------------------------------------------------------
def prob_magnitude(input: np.ndarray) -> float:
    return float(getattr(_lib, f"qip_oracle_prob_magnitude_{_suf(input)}")(input.ctypes.data, input.size))


def measure_prob(n, measured, indices, input: np.ndarray, input_offset: int = 0) -> float:
    return float(getattr(_lib, f"qip_oracle_measure_prob_{_suf(input)}")(
        n, measured, _u64(indices), len(indices), input.ctypes.data, input.size, input_offset))


def measure_probs(n, indices, input: np.ndarray, input_offset: int = 0) -> np.ndarray:
    real = np.float64 if input.dtype == np.complex128 else np.float32
    out = np.empty(1 << len(indices), dtype=real)
    getattr(_lib, f"qip_oracle_measure_probs_{_suf(input)}")(
        n, _u64(indices), len(indices), input.ctypes.data, input.size, input_offset, out.ctypes.data)
    return out


def soft_measure(n, indices, input: np.ndarray, rand_u01: float, input_offset: int = 0) -> int:
    return int(getattr(_lib, f"qip_oracle_soft_measure_{_suf(input)}")(
        n, _u64(indices), len(indices), input.ctypes.data, input.size, input_offset, rand_u01))


def measure_state(n, indices, measured: Tuple[int, float], input: np.ndarray, output: np.ndarray,
                  offsets: Tuple[int, int] = (0, 0)) -> bool:
    m, p = measured
    return bool(getattr(_lib, f"qip_oracle_measure_state_{_suf(input)}")(
        n, _u64(indices), len(indices), m, p, input.ctypes.data, input.size, output.ctypes.data, output.size,
        offsets[0], offsets[1]))


def measure(n, indices, input: np.ndarray, output: np.ndarray, forced: Optional[int] = None,
            rand_u01: float = 0.0) -> Tuple[int, float]:
    """measure (measurement_ops.rs:190-214)"""
    m = forced if forced is not None else soft_measure(n, indices, input, rand_u01)
    p = measure_prob(n, m, indices, input)
    measure_state(n, indices, (m, p), input, output)
    return m, p


# ---- run loop (qip/src/builder.rs:400-519), restated independently of rustqip_amd.builder ----
def lower(kind: str, indices: Sequence[int], param=None) -> Optional[MatrixOp]:
    """builder.rs:436-498.  Built with the raw MatrixOp constructors (ops.rs:49-91)."""
    idx = list(indices)
    one, zero, im = complex(1.0, 0.0), complex(0.0, 0.0), complex(0.0, 1.0)
    if kind == "X":
        return MatrixOp.new_matrix(idx, [zero, one, one, zero])
    if kind == "Y":
        return MatrixOp.new_matrix(idx, [zero, -im, im, zero])
    if kind == "Z":
        return MatrixOp.new_matrix(idx, [one, zero, zero, -one])
    if kind == "H":
        nl = complex(1.0 * math.sqrt(0.5), 0.0 * math.sqrt(0.5))  # Complex::one() * FRAC_1_SQRT_2 (:448-449)
        return MatrixOp.new_matrix(idx, [nl, nl, nl, -nl])
    if kind == "S":
        return MatrixOp.new_matrix(idx, [one, zero, zero, im])
    if kind == "T":
        return MatrixOp.new_matrix(idx, [one, zero, zero, cmath.rect(1.0, math.pi / 4)])
    if kind == "CNOT":
        return MatrixOp.new_control([idx[0]], idx[1:], MatrixOp.new_matrix(idx[1:], [zero, one, one, zero]))
    if kind == "MAT":
        return MatrixOp.new_matrix(idx, param)
    if kind == "SWAP":
        x = len(idx) // 2
        return MatrixOp.new_swap(idx[:x], idx[x:])
    if kind == "Rz":
        h_theta = float(param) * 0.5
        return MatrixOp.new_matrix(idx, [cmath.rect(1.0, -h_theta), zero, zero, cmath.rect(1.0, h_theta)])
    if kind == "GlobalPhase":
        return None
    raise ValueError(kind)


def run_pipeline(n: int, pipeline: Sequence[Tuple[Sequence[int], str, object]], initial_index: int,
                 forced_measurements: Sequence[int] = (), dtype=np.complex128):
    """calculate_state_with_init (builder.rs:400-519) for pipelines without mid-circuit
    GlobalPhase / stochastic stages (those trigger the reference's stale-buffer quirk, SURVEY.md
    App. C Q4, which neither this oracle nor the product reproduces: the buffers are only
    swapped after a stage that wrote the arena)."""
    state = np.zeros(1 << n, dtype=dtype)
    arena = np.zeros(1 << n, dtype=dtype)
    state[initial_index] = 1
    forced = list(forced_measurements)
    results: List[tuple] = []
    for indices, kind, param in pipeline:
        if kind == "Measurement":
            want = forced.pop(0) if forced else None
            if want is None:
                raise ValueError("oracle run loop needs forced measurement outcomes")
            m, p = measure(n, list(indices), state, arena, forced=want)
            results.append(("Single", m, p))
            if p != 0:
                state, arena = arena, state
        elif kind == "StochasticMeasurement":
            results.append(("Stochastic", measure_probs(n, list(indices), state), None))
        else:
            op = lower(kind, indices, param)
            if op is None:
                continue
            apply_op_overwrite(n, op, state, arena)
            state, arena = arena, state
    return state, results
