"""SURVEY §8 row a (apply_op / apply_op_overwrite / apply_op_row) for the OTHER instances of the kernel's generic element type `P`
(qip-iterators/src/matrix_ops.rs:98-107): real f64 / f32 (the reference's benches, benches/matmul_bench.rs) and integer i64 / i32
(its unit tests, matrix_ops.rs:271-374), on host slices (qip_hip_apply_op_host, qip_hip_apply_op_row_host) and on device slices
(qip_hip_apply_op_device).  Bar: bit-equal to the oracle for every op kind, window and type — one lane folds one row in the
reference's order, nothing is contracted.  The reference's own vectors in these types: tests/test_golden_fixtures.py."""
import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd.ops import MatrixOp
from test_oracle_golden import REAL_TYPES, random_real_ops, windows_accumulate

pytestmark = pytest.mark.gpu


class Buf:
    """device memory for a slice-level call without PyTorch in the test process: the amplitude buffer of a HipState (hipMalloc'd,
    16 bytes per amplitude) holding the bytes of a numpy array"""

    def __init__(self, arr):
        self.dtype, self.length, self.nbytes = arr.dtype, arr.size, arr.nbytes
        m = max(1, int(np.ceil(np.log2(max(arr.nbytes, 32) / 16))))
        self.st = q.HipState(m)
        raw = np.zeros((1 << m) * 16, dtype=np.uint8)
        raw[:arr.nbytes] = np.ascontiguousarray(arr).view(np.uint8).ravel()
        self.st.upload(raw.view(np.complex128))
        self.st.sync()

    def slice(self):
        return self.st.as_slice(self.dtype, 0, self.length)

    def get(self):
        device_sync()  # (the call launched on the null stream; the state downloads on its own)
        return self.st.download().view(np.uint8)[:self.nbytes].view(self.dtype).copy()

    def close(self):
        self.st.close()


def device_sync():
    import ctypes

    assert ctypes.CDLL("libamdhip64.so").hipDeviceSynchronize() == 0

# (whole vector; ragged windows; an empty input; the input's tail; an empty output; windows made of whole 16-byte vectors — the
#  literal kernel's V-rows-per-lane form; the same shifted by one element — back to one row per lane)
WINDOWS = lambda N: ((0, N, 0, N), (N // 8, N // 2 + 5, N // 16, N - N // 4), (0, 0, 3, 5), (N - 28, 28, 0, N), (0, N, N // 2, 0),  # noqa: E731
                     (N // 8, N // 2, N // 4, N // 2), (N // 8 + 1, N // 2, N // 4 + 1, N // 2))


def vector(rng, size, dtype):
    if np.issubdtype(dtype, np.integer):
        return rng.integers(-4, 5, size=size).astype(dtype)
    if np.issubdtype(dtype, np.complexfloating):
        return (rng.standard_normal(size) + 1j * rng.standard_normal(size)).astype(dtype)
    return rng.standard_normal(size).astype(dtype)


@pytest.mark.parametrize("dtype", REAL_TYPES)
def test_host_slices_every_kind_and_window(O, dtype):
    n = 9
    N = 1 << n
    rng = np.random.default_rng(11)
    integer = np.issubdtype(dtype, np.integer)
    for op in random_real_ops(n, rng, integer):
        x, y0 = vector(rng, N, dtype), vector(rng, N, dtype)
        for (io, il, oo, ol) in WINDOWS(N):
            xin = np.ascontiguousarray(x[io:io + il])
            for acc in (True, False):
                want, got = y0[:ol].copy(), y0[:ol].copy()
                O.apply_op(n, op, xin, want, io, oo, accumulate=acc)
                (q.apply_op if acc else q.apply_op_overwrite)(n, op, xin, got, io, oo)
                assert got.dtype == dtype and np.array_equal(got, want), (op, dtype, io, il, oo, ol, acc)
            if ol:
                r = int(rng.integers(0, ol))
                assert q.apply_op_row(n, op, xin, r, io, oo) == O.apply_op_row(n, op, xin, r, io, oo)


def test_integer_wraps_like_the_oracle():
    x = np.array([1 << 30, (1 << 31) - 1], dtype=np.int32)
    out = np.array([0, 1], dtype=np.int32)
    q.apply_op(1, MatrixOp.new_matrix([0], [4, 0, 0, 1]), x, out)
    assert list(out) == [0, -(1 << 31)]


@pytest.mark.parametrize("dtype", REAL_TYPES + (np.complex128, np.complex64))
def test_device_slices(O, dtype):
    """tensors stay on the GPU; the kernarg route (dense k <= 4, Swap, their Controls) and the uploaded-payload route (dense k = 5,
    SparseMatrix) against the oracle, windows included; complex P goes through the state path's literal kernel"""
    n = 12
    N = 1 << n
    rng = np.random.default_rng(12)
    integer = np.issubdtype(dtype, np.integer)
    real = not np.issubdtype(dtype, np.complexfloating)
    for op in random_real_ops(n, rng, integer):
        x, y0 = vector(rng, N, dtype), vector(rng, N, dtype)
        for (io, il, oo, ol) in WINDOWS(N)[:4] + WINDOWS(N)[5:]:
            xin = np.ascontiguousarray(x[io:io + il])
            for acc in (True, False):
                want = y0[:ol].copy()
                O.apply_op(n, op, xin, want, io, oo, accumulate=acc)
                d_in, d_out = Buf(xin), Buf(y0[:ol])
                q.apply_op_device(n, op, d_in.slice(), d_out.slice(), io, oo, accumulate=acc)
                got = d_out.get()
                d_in.close(), d_out.close()
                # real / integer P: one lane = the reference's fold (bit-equal); complex P: the literal kernel's bar of the state path
                assert np.array_equal(got, want) if real else np.max(np.abs(got - want), initial=0) <= (1e-12 if dtype == np.complex128 else 1e-5), (op, dtype, io, acc)


@pytest.mark.parametrize("dtype", REAL_TYPES)
def test_whole_vector_group_kernel_every_shape(O, dtype):
    """both windows = the whole vector, a dense op or Swap on <= 4 distinct indices: k_real_groups (each input read once, 16-byte
    accesses unless an index bit sits below them) — every (indices, controls, kind) shape, with the lowest index bit at position
    0, 1 and above, accumulate and overwrite, against the oracle AND against the literal kernel (global option force_generic)"""
    n = 8
    N = 1 << n
    rng = np.random.default_rng(21)
    integer = np.issubdtype(dtype, np.integer)
    vals = lambda c: rng.integers(-3, 4, size=c).astype(float) if integer else rng.standard_normal(c)  # noqa: E731
    shapes = []
    for low in (n - 1, n - 2, n - 4):  # qubit index of the lowest index bit: position 0, 1, 3
        others = [i for i in rng.permutation(n - 4)]
        for k_all in (1, 2, 3, 4):
            idx = [int(v) for v in others[:k_all - 1]] + [low]
            rng.shuffle(idx)
            for nc in range(k_all):
                inner = MatrixOp.new_matrix(idx[nc:], vals(4 ** (k_all - nc)))
                shapes.append(inner if nc == 0 else MatrixOp.new_control(idx[:nc], idx[nc:], inner))
                if (k_all - nc) % 2 == 0:
                    h = (k_all - nc) // 2
                    sw = MatrixOp.new_swap(idx[nc:nc + h], idx[nc + h:])
                    shapes.append(sw if nc == 0 else MatrixOp.new_control(idx[:nc], idx[nc:], sw))
    shapes.append(MatrixOp.new_matrix([2, 5], [1, 0, 2, 0, 0, 0, 0, 3, 0, 0, 0, 0, -1, 0, 0, 0]))  # zero entries, a zero row
    # both of the two lowest index bits in the op (a 4-byte P then keeps pairs), in either order, plain / controlled / swapped
    for a_, b_ in ((n - 1, n - 2), (n - 2, n - 1)):
        shapes += [MatrixOp.new_matrix([a_, b_], vals(16)), MatrixOp.new_swap([a_], [b_]), MatrixOp.new_matrix([a_, 1, b_], vals(64)),
                   MatrixOp.new_control([a_], [b_], MatrixOp.new_matrix([b_], vals(4))),
                   MatrixOp.new_control([2], [a_, b_], MatrixOp.new_swap([a_], [b_])),
                   MatrixOp.new_control([a_, 3], [b_], MatrixOp.new_matrix([b_], vals(4)))]
    for op in shapes:
        x, y0 = vector(rng, N, dtype), vector(rng, N, dtype)
        if not integer:
            x[3], x[N - 1] = -0.0, np.inf  # (0 + 1 * -0 = +0; a skipped zero entry never meets the infinity)
        for acc in (True, False):
            want = y0.copy()
            O.apply_op(n, op, x, want, accumulate=acc)
            outs = []
            for generic in (0, 1):
                q.set_global_option("force_generic", generic)
                try:
                    d_in, d_out = Buf(x), Buf(y0)
                    q.apply_op_device(n, op, d_in.slice(), d_out.slice(), accumulate=acc)
                    outs.append(d_out.get())
                    d_in.close(), d_out.close()
                finally:
                    q.set_global_option("force_generic", 0)
            for got in outs:
                assert np.array_equal(got, want, equal_nan=True) and np.array_equal(np.signbit(got), np.signbit(want)) if not integer else np.array_equal(got, want), (op, dtype, acc)


@pytest.mark.parametrize("dtype", (np.int64, np.int32))
def test_input_windows_accumulate_to_the_whole_vector(O, dtype):
    """the reference's provision for several devices (offset windows, matrix_ops.rs:96-97) through the HIP path: apply_op
    accumulated over a partition of the input into four windows, into each of two output windows, is the whole product —
    exactly, in integer arithmetic — and equals the oracle's whole-vector result"""
    n = 10
    rng = np.random.default_rng(18)
    for op in random_real_ops(n, rng, True):
        x = rng.integers(-(1 << 20), 1 << 20, size=1 << n).astype(dtype)
        want = np.zeros(1 << n, dtype=dtype)
        O.apply_op(n, op, x, want)
        assert np.array_equal(windows_accumulate(q.apply_op, n, op, x, dtype), want), op


def test_reference_bench_shape_ones(O):
    """qip-iterators/benches/matmul_bench.rs:19-33 (n = 12) and :163-177 (n = 20): P = f64, a 2 x 2 matrix of ones on qubit 0,
    ones in, apply_op accumulating into the same output call after call"""
    op = MatrixOp.new_matrix([0], [1.0, 1.0, 1.0, 1.0])
    for n in (12, 20):
        x = np.ones(1 << n)
        want = np.zeros(1 << n)
        d_in, d_out = Buf(x), Buf(want)
        for _ in range(3):
            O.apply_op(n, op, x, want)
            q.apply_op_device(n, op, d_in.slice(), d_out.slice())
        assert np.array_equal(d_out.get(), want) and want[0] == 6.0


@pytest.mark.slow
@pytest.mark.parametrize("dtype", (np.float64, np.float32))
def test_real_vector_of_2_to_26(O, dtype):
    """a vector the size of HBM traffic (512 / 256 MiB): every row against the oracle"""
    n = 26
    rng = np.random.default_rng(13)
    x = rng.standard_normal(1 << n).astype(dtype)
    d_in, d_out = Buf(x), Buf(np.zeros(1 << n, dtype=dtype))
    for op in (MatrixOp.new_matrix([0], rng.standard_normal(4)), MatrixOp.new_matrix([25, 3], rng.standard_normal(16)),
               MatrixOp.new_control([7], [20], MatrixOp.new_matrix([20], rng.standard_normal(4))), MatrixOp.new_swap([1, 24], [13, 2])):
        want = np.zeros(1 << n, dtype=dtype)
        O.apply_op_overwrite(n, op, x, want)
        q.apply_op_device(n, op, d_in.slice(), d_out.slice(), accumulate=False)
        assert np.array_equal(d_out.get(), want), op


def test_argument_errors():
    x, y, z = Buf(np.ones(4)), Buf(np.zeros(4)), Buf(np.zeros(4, dtype=np.float32))
    ident = MatrixOp.new_matrix([0], [1, 0, 0, 1])
    with pytest.raises(q.CircuitError, match="alias"):
        q.apply_op_device(2, ident, x.slice(), x.slice())
    with pytest.raises(q.CircuitError, match="dtype"):
        q.apply_op_device(2, ident, x.slice(), z.slice())
    with pytest.raises(q.CircuitError, match="out of range"):
        q.apply_op_device(2, MatrixOp.new_matrix([5], [1, 0, 0, 1]), x.slice(), y.slice())
    with pytest.raises(q.CircuitError, match="imaginary"):
        q.apply_op_device(2, MatrixOp.new_matrix([0], [1j, 0, 0, 1]), x.slice(), y.slice())
    with pytest.raises(q.CircuitError, match="dtype"):
        q.HipState(3, dtype=np.float64)  # a state is Complex<P>
    q.apply_op_device(2, ident, x.slice(), y.slice())  # (and the handles are still good)
    assert list(y.get()) == [1, 1, 1, 1]
