"""apply_op / apply_op_overwrite (host-pointer twins) and the device-resident state.

Mirrors, for the hot path only:
  qip_iterators::matrix_ops::apply_op            (qip-iterators/src/matrix_ops.rs:98-123)
  qip_iterators::matrix_ops::apply_op_overwrite  (:127-152)
  the state / arena pair of LocalBuilder::calculate_state_with_init (qip/src/builder.rs:400-519)
  qip::state_ops::measurement_ops                (qip/src/state_ops/measurement_ops.rs)
Every function calls straight into libqip_hip.so (HIP kernels); nothing is computed on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .ops import CircuitError, MatrixOp, complex_dtype


class QipHipError(RuntimeError):
    """A non-zero status from the C ABI that is not a descriptor-validation error."""


def _check(rc: int) -> None:
    if rc == _ffi.QIP_OK:
        return
    msg = _ffi.last_error()
    if rc == _ffi.QIP_ERR_INVALID:
        raise CircuitError(msg)
    raise QipHipError(f"qip_hip status {rc}: {msg}")


def device_count() -> int:
    return int(_ffi.lib.qip_hip_device_count())


def set_global_option(key: str, value: int) -> None:
    _check(_ffi.lib.qip_hip_set_global_option(key.encode(), int(value)))


def _dtype_of(arr: np.ndarray) -> int:
    if arr.dtype == np.complex128:
        return _ffi.QIP_C64
    if arr.dtype == np.complex64:
        return _ffi.QIP_C32
    raise CircuitError(f"unsupported amplitude dtype {arr.dtype}; use complex128 or complex64")


_SLICE_DTYPES = {np.dtype(np.complex128): _ffi.QIP_C64, np.dtype(np.complex64): _ffi.QIP_C32, np.dtype(np.float64): _ffi.QIP_F64,
                 np.dtype(np.float32): _ffi.QIP_F32, np.dtype(np.int64): _ffi.QIP_I64, np.dtype(np.int32): _ffi.QIP_I32}


def _slice_dtype_of(arr) -> int:
    """element type `P` of a slice-level call (apply_op<P> is generic, matrix_ops.rs:98-107): complex, real or integer"""
    try:
        return _SLICE_DTYPES[np.dtype(arr.dtype)]
    except (KeyError, TypeError):
        raise CircuitError(f"unsupported element dtype {arr.dtype}; use complex128/64, float64/32 or int64/32") from None


def _apply_host(n, op, input, output, input_offset, output_offset, accumulate):
    if not (isinstance(input, np.ndarray) and isinstance(output, np.ndarray)):
        raise TypeError("input and output must be numpy arrays")
    dtype = _slice_dtype_of(output)
    if _slice_dtype_of(input) != dtype:
        raise CircuitError("input and output precision differ")
    if not (input.flags.c_contiguous and output.flags.c_contiguous and output.flags.writeable):
        raise CircuitError("input/output must be contiguous, output writeable")
    if np.shares_memory(input, output):
        raise CircuitError("input and output must not alias (&[P] vs &mut [P])")
    cop = op.to_c(dtype)
    _check(
        _ffi.lib.qip_hip_apply_op_host(
            dtype, n, C.byref(cop), input.ctypes.data, input.size, output.ctypes.data, output.size,
            int(input_offset), int(output_offset), int(accumulate),
        )
    )


def apply_op(n: int, op: MatrixOp, input: np.ndarray, output: np.ndarray, input_offset: int = 0,
             output_offset: int = 0) -> None:
    """output[r] += (op · input)[r]   — matrix_ops.rs:98-123, same argument order."""
    _apply_host(n, op, input, output, input_offset, output_offset, True)


def apply_op_overwrite(n: int, op: MatrixOp, input: np.ndarray, output: np.ndarray, input_offset: int = 0,
                       output_offset: int = 0) -> None:
    """output[r] = (op · input)[r]   — matrix_ops.rs:127-152."""
    _apply_host(n, op, input, output, input_offset, output_offset, False)


def apply_op_row(n: int, op: MatrixOp, input: np.ndarray, outputrow: int, input_offset: int = 0,
                 output_offset: int = 0):
    """apply_op_row (matrix_ops.rs:38-59): the value of row output_offset + outputrow (a `P`: complex, float or int)."""
    dtype = _slice_dtype_of(input)
    cop = op.to_c(dtype)
    out = np.zeros(1, dtype=input.dtype)
    _check(_ffi.lib.qip_hip_apply_op_row_host(dtype, n, C.byref(cop), input.ctypes.data, input.size, int(outputrow),
                                              int(input_offset), int(output_offset), out.ctypes.data))
    return out[0].item()


class DeviceSlice:
    """`&[P]` / `&mut [P]` in device memory: a raw pointer to `length` elements of numpy type `dtype` on GPU `device`
    (from hipMalloc, a torch tensor's data_ptr(), HipState.as_slice(...))"""

    def __init__(self, ptr: int, length: int, dtype, device: int = 0):
        self.ptr, self.length, self.dtype, self.device = int(ptr), int(length), np.dtype(dtype), int(device)


def _as_slice(t) -> DeviceSlice:
    if isinstance(t, DeviceSlice):
        return t
    dt = _TORCH_DTYPES.get(str(t.dtype))  # a torch tensor (device memory and streams are torch's job; no torch types in the C ABI)
    if dt is None or not (t.is_cuda and t.is_contiguous()):
        raise CircuitError(f"unsupported element dtype {t.dtype}, or not a contiguous tensor on a GPU")
    return DeviceSlice(t.data_ptr() if t.numel() else 0, t.numel(), dt, t.device.index or 0)


_TORCH_DTYPES = {"torch.complex128": np.complex128, "torch.complex64": np.complex64, "torch.float64": np.float64,
                 "torch.float32": np.float32, "torch.int64": np.int64, "torch.int32": np.int32}


def apply_op_device(n: int, op: MatrixOp, input, output, input_offset: int = 0, output_offset: int = 0, *,
                    accumulate: bool = True, stream: int = 0) -> None:
    """apply_op (accumulate) / apply_op_overwrite on DEVICE slices for any `P` (qip_hip_apply_op_device): `input` / `output`
    are DeviceSlice objects or contiguous 1-D torch tensors on one GPU, `stream` a raw hipStream_t (0 = null stream).  Nothing
    is copied to the host; with a dense op on k <= 4 qubits or a Swap the call only launches.  `op` may be a MatrixOp or the
    descriptor `op.to_c(dtype)` built once (the reference's benches build their op once, outside the timed loop)."""
    i, o = _as_slice(input), _as_slice(output)
    if i.dtype != o.dtype or i.device != o.device:
        raise CircuitError(f"input and output differ in element dtype or device ({i.dtype} on {i.device}, {o.dtype} on {o.device})")
    dtype = _slice_dtype_of(o)
    eb = o.dtype.itemsize
    if i.length and o.length and i.ptr < o.ptr + o.length * eb and o.ptr < i.ptr + i.length * eb:
        raise CircuitError("input and output must not alias (&[P] vs &mut [P])")
    cop = op if isinstance(op, _ffi.QipOp) else op.to_c(dtype)
    _check(_ffi.lib.qip_hip_apply_op_device(dtype, o.device, stream or None, n, C.byref(cop), i.ptr or None, i.length,
                                            o.ptr or None, o.length, int(input_offset), int(output_offset), int(accumulate)))


def measure_probs(n: int, indices: Sequence[int], input: np.ndarray, input_offset: int = 0) -> np.ndarray:
    """measure_probs (measurement_ops.rs:115-127) on a host window `input` = amplitudes [input_offset, +len)."""
    out = np.empty(1 << len(indices), dtype=np.float64)
    _check(_ffi.lib.qip_hip_measure_probs_host(_dtype_of(input), n, _u64_array(indices), len(indices), input.ctypes.data,
                                               input.size, int(input_offset), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def measure_prob(n: int, measured: int, indices: Sequence[int], input: np.ndarray, input_offset: int = 0) -> float:
    """measure_prob (measurement_ops.rs:44-58) on a host window."""
    out = C.c_double()
    _check(_ffi.lib.qip_hip_measure_prob_host(_dtype_of(input), n, int(measured), _u64_array(indices), len(indices),
                                              input.ctypes.data, input.size, int(input_offset), C.byref(out)))
    return out.value


def make_op_matrix(n: int, op: MatrixOp, dtype=np.complex128) -> np.ndarray:  # (any `P`: the reference's tests use i32)
    """Full 2^n x 2^n matrix of `op`, column by column through apply_op on basis vectors —
    the reference's test/debug helper (qip/src/state_ops/matrix_ops.rs:246-257,
    qip-iterators/src/matrix_ops.rs:229-255).  Returns M with M[r, c] = <r|op|c>."""
    N = 1 << n
    cols = []
    for i in range(N):
        inp = np.zeros(N, dtype=dtype)
        out = np.zeros(N, dtype=dtype)
        inp[i] = 1
        apply_op(n, op, inp, out, 0, 0)
        cols.append(out)
    return np.stack(cols, axis=1)


def _u64_array(values: Sequence[int]):
    arr = (C.c_uint64 * len(values))(*[int(v) for v in values])
    return arr


class HipState:
    """2^n amplitudes resident in HBM; gates are applied in place by HIP kernels.

    Replaces the `state` / `arena` Vec pair of the reference run loop (builder.rs:406-407,514).
    """

    def __init__(self, n: int, dtype=np.complex128, device: int = 0, *, wrap_ptr: Optional[int] = None,
                 scratch_ptr: Optional[int] = None, stream: Optional[int] = None):
        self.n = int(n)
        self.device = int(device)
        self.np_dtype = np.dtype(dtype)
        self.dtype = _dtype_of(np.empty(0, dtype=dtype))
        self._h = C.c_void_p()
        if wrap_ptr is None:
            _check(_ffi.lib.qip_hip_state_create(self.n, self.dtype, device, C.byref(self._h)))
        else:
            _check(
                _ffi.lib.qip_hip_state_wrap(self.n, self.dtype, device, C.c_void_p(wrap_ptr),
                                            C.c_void_p(scratch_ptr or 0), C.c_void_p(stream or 0),
                                            C.byref(self._h))
            )

    @classmethod
    def from_handle(cls, handle, n: int, dtype=np.complex128) -> "HipState":
        """A view of a qip_hip_state somebody else owns (the shard of a qip_hip_dist): never destroyed from here."""
        self = cls.__new__(cls)
        self.n = int(n)
        self.np_dtype = np.dtype(dtype)
        self.dtype = _dtype_of(np.empty(0, dtype=dtype))
        self._h = C.c_void_p(handle.value if hasattr(handle, "value") else int(handle))
        self._borrowed = True
        return self

    # -- lifetime --------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            if not getattr(self, "_borrowed", False):
                _ffi.lib.qip_hip_state_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self) -> int:
        return 1 << self.n

    # -- data movement -----------------------------------------------------------------
    def init_basis(self, index: int) -> None:
        _check(_ffi.lib.qip_hip_state_init_basis(self._h, int(index)))

    def upload(self, amps: np.ndarray, offset: int = 0) -> None:
        amps = np.ascontiguousarray(amps, dtype=self.np_dtype)
        _check(_ffi.lib.qip_hip_state_upload(self._h, amps.ctypes.data, int(offset), amps.size))

    def download(self, offset: int = 0, length: Optional[int] = None) -> np.ndarray:
        length = (1 << self.n) - offset if length is None else int(length)
        out = np.empty(length, dtype=self.np_dtype)
        _check(_ffi.lib.qip_hip_state_download(self._h, out.ctypes.data, int(offset), length))
        return out

    def download_indices(self, indices) -> np.ndarray:
        """amplitudes at an explicit list of indices (one gather kernel; the parity checks' scattered sub-cubes)"""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = np.empty(idx.size, dtype=self.np_dtype)
        _check(_ffi.lib.qip_hip_state_download_indices(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), idx.size, out.ctypes.data))
        return out

    def device_ptr(self) -> int:
        p = C.c_void_p()
        _check(_ffi.lib.qip_hip_state_device_ptr(self._h, C.byref(p)))
        return int(p.value or 0)

    def as_slice(self, dtype=None, offset: int = 0, length: Optional[int] = None) -> DeviceSlice:
        """the amplitude buffer seen as elements of `dtype` (default: the state's own), [offset, offset + length) of them:
        the device memory of a slice-level call (apply_op_device)"""
        dt = np.dtype(dtype or self.np_dtype)
        total = (len(self) * self.np_dtype.itemsize) // dt.itemsize
        length = total - offset if length is None else length
        if offset < 0 or length < 0 or offset + length > total:
            raise CircuitError("slice outside the state's buffer")
        return DeviceSlice(self.device_ptr() + offset * dt.itemsize, length, dt, self.device)

    def scratch_ptr(self) -> int:
        p = C.c_void_p()
        _check(_ffi.lib.qip_hip_state_scratch_ptr(self._h, C.byref(p)))
        return int(p.value or 0)

    def swap_buffers(self) -> None:
        """make the scratch buffer current (after an external out-of-place step wrote it)"""
        _check(_ffi.lib.qip_hip_state_swap_buffers(self._h))

    def sync(self) -> None:
        _check(_ffi.lib.qip_hip_state_sync(self._h))

    def copy_from(self, other: "HipState") -> None:
        """self <- other, device to device (same n, precision and device)"""
        _check(_ffi.lib.qip_hip_state_copy_from(self._h, other._h))

    def max_abs_diff(self, other: "HipState") -> Tuple[float, int]:
        """(max_i |self_i - other_i|, number of amplitudes that are not IEEE-equal) over the whole vector"""
        worst, differ = C.c_double(), C.c_uint64()
        _check(_ffi.lib.qip_hip_state_max_abs_diff(self._h, other._h, C.byref(worst), C.byref(differ)))
        return worst.value, int(differ.value)

    def set_option(self, key: str, value: int) -> None:
        _check(_ffi.lib.qip_hip_state_set_option(self._h, key.encode(), int(value)))

    # -- gates ---------------------------------------------------------------------------
    def permute_bits(self, pi) -> None:
        """new[j] = old[src(j)], bit pi[d] of src(j) = bit d of j: any permutation of the index bits in one sweep"""
        if len(pi) != self.n:
            raise CircuitError("the permutation must list all n index bits")
        arr = (C.c_uint32 * self.n)(*[int(b) for b in pi])
        _check(_ffi.lib.qip_hip_state_permute_bits(self._h, arr))

    def apply_op(self, op: MatrixOp) -> None:
        """state <- op · state  (apply_op_overwrite + swap, builder.rs:499,514)."""
        cop = op.to_c(self.dtype)
        _check(_ffi.lib.qip_hip_state_apply_op(self._h, C.byref(cop)))

    def compile_ops(self, ops: Iterable[MatrixOp]):
        """Marshal a circuit once; returns an opaque object for apply_compiled."""
        cops = [op.to_c(self.dtype) for op in ops]
        arr = (_ffi.QipOp * len(cops))(*cops)
        return (arr, cops)  # cops keeps the buffers alive

    def apply_compiled(self, compiled) -> None:
        arr, _keep = compiled
        _check(_ffi.lib.qip_hip_state_apply_ops(self._h, arr, len(arr)))

    def apply_ops(self, ops: Iterable[MatrixOp]) -> None:
        self.apply_compiled(self.compile_ops(list(ops)))

    def compile_program(self, ops: Iterable[MatrixOp]) -> "HipProgram":
        """Record a circuit once; HipProgram.run() replays it as ONE hipGraph launch (launch-bound small
        states), falling back to eager application when a graph is not possible."""
        return HipProgram(self, list(ops))

    # -- measurement (measurement_ops.rs) ------------------------------------------------
    def norm_sqr(self) -> float:
        out = C.c_double()
        _check(_ffi.lib.qip_hip_state_norm_sqr(self._h, C.byref(out)))
        return out.value

    def measure_probs(self, indices: Sequence[int]) -> np.ndarray:
        k = len(indices)
        out = np.empty(1 << k, dtype=np.float64)
        _check(_ffi.lib.qip_hip_state_measure_probs(self._h, _u64_array(indices), k,
                                                    out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def measure_prob(self, measured: int, indices: Sequence[int]) -> float:
        out = C.c_double()
        _check(_ffi.lib.qip_hip_state_measure_prob(self._h, int(measured), _u64_array(indices), len(indices),
                                                   C.byref(out)))
        return out.value

    def soft_measure(self, indices: Sequence[int], rand_u01: float) -> int:
        out = C.c_uint64()
        _check(_ffi.lib.qip_hip_state_soft_measure(self._h, _u64_array(indices), len(indices), float(rand_u01),
                                                   C.byref(out)))
        return int(out.value)

    def measure(self, indices: Sequence[int], measured: Optional[int] = None,
                rand_u01: float = 0.0) -> Tuple[int, float]:
        m = C.c_uint64()
        p = C.c_double()
        forced = -1 if measured is None else int(measured)
        _check(_ffi.lib.qip_hip_state_measure(self._h, _u64_array(indices), len(indices), forced,
                                              float(rand_u01), C.byref(m), C.byref(p)))
        return int(m.value), p.value

    def measure_state(self, indices: Sequence[int], measured: int, prob: float) -> None:
        """measure_state (measurement_ops.rs:220-269) with a caller-supplied probability."""
        _check(_ffi.lib.qip_hip_state_measure_state(self._h, _u64_array(indices), len(indices), int(measured),
                                                    float(prob)))

    # -- profiling -----------------------------------------------------------------------
    def profile(self) -> dict:
        """Per-kernel-class {launches, total_ms, algorithmic_bytes} gathered while option
        'profile' = 1 (HIP events on the handle's stream)."""
        out = {}
        for cls in range(_ffi.lib.qip_hip_kernel_class_count()):
            launches, ms, by = C.c_uint64(), C.c_double(), C.c_double()
            _check(_ffi.lib.qip_hip_state_profile_get(self._h, cls, C.byref(launches), C.byref(ms), C.byref(by)))
            if launches.value:
                out[_ffi.lib.qip_hip_kernel_class_name(cls).decode()] = {
                    "launches": int(launches.value), "total_ms": ms.value, "algorithmic_bytes": by.value,
                }
        return out

    def profile_reset(self) -> None:
        _check(_ffi.lib.qip_hip_state_profile_reset(self._h))


class HipProgram:
    """qip_hip_program: a circuit captured into a hipGraph bound to a HipState."""

    def __init__(self, state: HipState, ops: List[MatrixOp]):
        self.state = state
        self._compiled = state.compile_ops(ops)  # keeps the descriptors alive for the program's lifetime
        self._p = C.c_void_p()
        arr, _ = self._compiled
        _check(_ffi.lib.qip_hip_program_create(state._h, arr, len(arr), C.byref(self._p)))

    def run(self) -> None:
        _check(_ffi.lib.qip_hip_program_run(self._p))

    @property
    def is_graph(self) -> bool:
        return bool(_ffi.lib.qip_hip_program_is_graph(self._p))

    def close(self) -> None:
        if self._p.value:
            _ffi.lib.qip_hip_program_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
