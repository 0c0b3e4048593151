"""Pin the CPU oracle against every golden vector the reference's own tests hold for the
hot path.  CPU only."""
import math

import numpy as np
import pytest

from oracle import qip_oracle as O
from rustqip_amd.ops import MatrixOp


def kron_helper(before, mat, after):
    """ndarray_kron_helper (qip-iterators/src/matrix_ops.rs:257-269)"""
    eye = np.eye(2)
    for _ in range(before):
        mat = np.kron(eye, mat)
    for _ in range(after):
        mat = np.kron(mat, eye)
    return mat


# ---- B1: qip-iterators/src/matrix_ops.rs:271-348 ----------------------------------------
@pytest.mark.parametrize("data,q", [([1, 0, 0, 1], 0), ([0, 1, 1, 0], 0), ([0, 1, 1, 0], 1), ([0, 1, 1, 0], 2),
                                     ([1, 2, 3, 4], 0)])
def test_b1_single_qubit_kron(data, q):
    n = 3
    op = MatrixOp.new_matrix([q], data)
    mat = O.make_op_matrix(n, op)
    comp = kron_helper(q, np.array(data, dtype=float).reshape(2, 2), n - 1 - q)
    assert np.array_equal(mat, comp.astype(np.complex128))


def test_b1_flip_mid_twobody():
    n = 4
    data = [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1]
    op = MatrixOp.new_matrix([1, 2], data)
    mat = O.make_op_matrix(n, op)
    comp = kron_helper(1, np.array(data, dtype=float).reshape(4, 4), 1)
    assert np.array_equal(mat, comp.astype(np.complex128))


# ---- B2: matrix_ops.rs:350-374 -------------------------------------------------------------
def test_b2_counting_order():
    data = list(range(16))
    comp = np.array(data, dtype=np.complex128).reshape(4, 4)
    assert np.array_equal(O.make_op_matrix(2, MatrixOp.new_matrix([0, 1], data)), comp)
    assert not np.array_equal(O.make_op_matrix(2, MatrixOp.new_matrix([1, 0], data)), comp)


# ---- B3: iterators/qubit_iterators.rs:289-379 (row -> nonzero column patterns) ---------------
def _row_cols(n, op):
    m = O.make_op_matrix(n, op)
    return [list(np.nonzero(m[r])[0]) for r in range(1 << n)], m


def test_b3_iterator_patterns():
    cols, _ = _row_cols(1, MatrixOp.new_matrix([0], [0, 1, 1, 0]))
    assert cols == [[1], [0]]
    cols, _ = _row_cols(1, MatrixOp.new_sparse([0], [[(1, 1)], [(0, 1)]]))
    assert cols == [[1], [0]]
    cols, m = _row_cols(2, MatrixOp.new_swap([0], [1]))
    assert cols == [[0], [2], [1], [3]] and np.all(m[m != 0] == 1)
    cnot = MatrixOp.new_control([0], [1], MatrixOp.new_matrix([1], [0, 1, 1, 0]))
    cols, m = _row_cols(2, cnot)
    assert cols == [[0], [1], [3], [2]] and np.all(m[m != 0] == 1)


# ---- B4: qip/src/state_ops/matrix_ops.rs:306-344 -----------------------------------------------
def test_b4_apply_identity_and_flip():
    inp = np.array([1, 0], dtype=np.complex128)
    out = np.zeros(2, dtype=np.complex128)
    O.apply_op(1, MatrixOp.new_matrix([0], [1, 0, 0, 1]), inp, out)
    assert np.array_equal(inp, out)
    out = np.zeros(2, dtype=np.complex128)
    O.apply_op(1, MatrixOp.new_matrix([0], [0, 1, 1, 0]), inp, out)
    assert np.array_equal(out, inp[::-1])


def test_b4_apply_swap_mat_first():
    inp = np.array([1, 0, 0, 0], dtype=np.complex128)
    out = np.zeros(4, dtype=np.complex128)
    O.apply_op(2, MatrixOp.new_matrix([0], [0, 1, 1, 0]), inp, out)
    assert np.array_equal(out, np.array([0, 0, 1, 0], dtype=np.complex128))
    out = np.zeros(4, dtype=np.complex128)
    O.apply_op(2, MatrixOp.new_matrix([1], [0, 1, 1, 0]), inp, out)
    assert np.array_equal(out, np.array([0, 1, 0, 0], dtype=np.complex128))


# ---- B6: measurement_ops.rs doctests :24-43, :136-152 and tests :290-335 --------------------------
def test_b6_measure_prob_doctest():
    inp = np.array([0, 0, 1, 0], dtype=np.complex128)
    assert O.measure_prob(2, 0, [0], inp) == 0.0
    assert O.measure_prob(2, 1, [0], inp) == 1.0
    assert O.measure_prob(2, 1, [0, 1], inp) == 1.0
    assert O.measure_prob(2, 2, [1, 0], inp) == 1.0


@pytest.mark.parametrize("r", [1e-12, 0.3, 0.999])  # rand::random::<f64>() in (0,1); r == 0.0 exactly picks index 0
def test_b6_soft_measure_doctest(r):
    inp = np.array([0, 0, 1, 0], dtype=np.complex128)
    assert O.soft_measure(2, [0], inp, r) == 1
    assert O.soft_measure(2, [1], inp, r) == 0
    assert O.soft_measure(2, [0, 1], inp, r) == 0b01
    assert O.soft_measure(2, [1, 0], inp, r) == 0b10


@pytest.mark.parametrize("m,expected", [(0, [math.sqrt(0.5), math.sqrt(0.5), 0, 0]),
                                        (1, [0, 0, math.sqrt(0.5), math.sqrt(0.5)])])
def test_b6_measure_state(m, expected):
    inp = np.full(4, 0.5, dtype=np.complex128)
    p = O.measure_prob(2, m, [0], inp)
    assert abs(p - 0.5) < np.finfo(float).eps
    out = inp.copy()
    assert O.measure_state(2, [0], (m, p), inp, out)
    assert np.array_equal(np.round(out * 1e10) / 1e10, np.round(np.array(expected, dtype=np.complex128) * 1e10) / 1e10)


def test_b6_measure_probs():
    inp = np.full(4, 0.5, dtype=np.complex128)
    assert list(O.measure_probs(2, [1], inp)) == [0.5, 0.5]


def test_measure_state_zero_prob_is_noop():
    inp = np.array([1, 0, 0, 0], dtype=np.complex128)
    out = np.full(4, 7.0, dtype=np.complex128)
    assert not O.measure_state(2, [0], (1, 0.0), inp, out)
    assert np.all(out == 7.0)  # measurement_ops.rs:230


# ---- B8: bit-util doctests -------------------------------------------------------------------------
def test_b8_bit_utils():
    assert O.flip_bits(3, 0b100) == 0b001
    assert O.flip_bits(3, 0b010) == 0b010
    assert O.flip_bits(4, 0b1010) == 0b0101
    assert O.set_bit(0, 1, True) == 2
    assert O.set_bit(1, 1, True) == 3
    assert O.set_bit(1, 0, False) == 0
    assert O.get_bit(2, 1) is True
    assert not O.get_bit(1, 1) and O.get_bit(1, 0)  # state_ops/matrix_ops.rs:264-274
    assert O.set_bit(1, 0, True) == 1
    assert O.entwine_bits(3, 0b010, 0b01, 0b1) == 0b011  # qip/src/utils.rs:13-20
    assert O.extract_bits(0b1010, [3, 0]) == 0b01  # :49-53
    assert O.get_flat_index(2, 1, 3) == 7


# ---- B7: README CSWAP (README.md:26-63), derived known answer -------------------------------------
def test_b7_cswap_known_answer():
    from rustqip_amd.builder import HipBuilder

    b = HipBuilder()
    q = b.qubit()
    ra = b.register(3)
    rb = b.register(3)
    q = b.h(q)
    cb = b.condition_with(q)
    ra, rb = cb.swap(ra, rb)
    q = cb.dissolve()
    q = b.h(q)
    unitary = [(e.indices, e.kind, e.param) for e in b.pipeline]
    assert len(unitary) == 191  # 1 H + 3 pairs x 3 CNOT x 21 + 1 H  
    assert b.n() == 7
    init = b.initial_index([(ra, 0b000), (rb, 0b001)])
    assert init == 4
    state, _ = O.run_pipeline(7, unitary, init)
    expect = np.zeros(128, dtype=np.complex128)
    expect[[4, 32, 68]] = 0.5
    expect[96] = -0.5
    assert np.max(np.abs(state - expect)) < 1e-12
    assert abs(O.prob_magnitude(state) - 1) < 1e-12
    probs = O.measure_probs(7, [0], state)
    assert np.allclose(probs, [0.5, 0.5], atol=1e-12)


# ---- beyond the reference's vectors: oracle vs an independent dense construction -------------------
def dense_of(n, indices, sub):
    """Full matrix of a k-qubit operator `sub` (2^k x 2^k, indices[0] = MSB) by explicit bit maps."""
    N, k = 1 << n, len(indices)
    full = np.zeros((N, N), dtype=np.complex128)
    pos = [n - 1 - q for q in indices]
    for r in range(N):
        mr = sum(((r >> pos[j]) & 1) << (k - 1 - j) for j in range(k))
        for mc in range(1 << k):
            c = r
            for j in range(k):
                bit = (mc >> (k - 1 - j)) & 1
                c = (c & ~(1 << pos[j])) | (bit << pos[j])
            full[r, c] = sub[mr, mc]
    return full


def controlled(nc, u):
    k = int(math.log2(u.shape[0]))
    m = np.eye(1 << (nc + k), dtype=np.complex128)
    m[-(1 << k):, -(1 << k):] = u
    return m


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    q, _ = np.linalg.qr(a)
    return q


@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_dense_construction(seed):
    rng = np.random.default_rng(seed)
    n = 6
    perm = list(rng.permutation(n))
    # dense 2-qubit
    u = rand_unitary(2, rng)
    assert np.allclose(O.make_op_matrix(n, MatrixOp.new_matrix(perm[:2], u.ravel())), dense_of(n, perm[:2], u), atol=1e-14)
    # Control with 2 controls over a 1-qubit op
    u1 = rand_unitary(1, rng)
    op = MatrixOp.new_control(perm[:2], perm[2:3], MatrixOp.new_matrix(perm[2:3], u1.ravel()))
    assert np.allclose(O.make_op_matrix(n, op), dense_of(n, perm[:3], controlled(2, u1)), atol=1e-14)
    # nested Control == flattened Control
    nested = MatrixOp.new_control(perm[:1], perm[1:3], MatrixOp.new_control(perm[1:2], perm[2:3], MatrixOp.new_matrix(perm[2:3], u1.ravel())))
    assert np.array_equal(O.make_op_matrix(n, nested), O.make_op_matrix(n, op))
    # Swap with h = 2
    a, b = perm[:2], perm[2:4]
    sw = np.zeros((16, 16))
    for m in range(16):
        sw[m, ((m & 3) << 2) | (m >> 2)] = 1
    assert np.array_equal(O.make_op_matrix(n, MatrixOp.new_swap(a, b)), dense_of(n, a + b, sw).astype(np.complex128))
    # SparseMatrix == the same data as a dense Matrix
    dense = rand_unitary(2, rng)
    dense[np.abs(dense) < 0.4] = 0
    for r in range(4):
        if not dense[r].any():
            dense[r, r] = 1
    rows = [[(c, dense[r, c]) for c in range(4) if dense[r, c] != 0] for r in range(4)]
    assert np.allclose(O.make_op_matrix(n, MatrixOp.new_sparse(perm[:2], rows)),
                       O.make_op_matrix(n, MatrixOp.new_matrix(perm[:2], dense.ravel())), atol=1e-15)


def test_oracle_window_identity():
    """rows of shard r = sum over input windows w of apply_op(n, op, in_w, out_r, w*S, r*S)."""
    rng = np.random.default_rng(7)
    n, shards = 6, 4
    S = (1 << n) // shards
    x = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(np.complex128)
    op = MatrixOp.new_matrix([0, 4], rand_unitary(2, rng).ravel())
    full = np.zeros(1 << n, dtype=np.complex128)
    O.apply_op(n, op, x, full)
    for r in range(shards):
        out = np.zeros(S, dtype=np.complex128)
        for w in range(shards):
            O.apply_op(n, op, x[w * S:(w + 1) * S].copy(), out, w * S, r * S)
        assert np.allclose(out, full[r * S:(r + 1) * S], atol=1e-14)


def test_oracle_f32_matches_f64_loosely():
    rng = np.random.default_rng(3)
    n = 5
    x = (rng.standard_normal(32) + 1j * rng.standard_normal(32))
    op = MatrixOp.new_matrix([2], rand_unitary(1, rng).ravel())
    o64 = np.zeros(32, dtype=np.complex128)
    o32 = np.zeros(32, dtype=np.complex64)
    O.apply_op(n, op, x.astype(np.complex128), o64)
    O.apply_op(n, op, x.astype(np.complex64), o32)
    assert np.allclose(o32, o64, atol=1e-5)


# ---- generic P: a real / integer vector (matrix_ops.rs:98-107; the reference's own B1 / B2 tests run on i32) ----------------------
REAL_TYPES = (np.float64, np.float32, np.int64, np.int32)


def random_real_ops(n, rng, integer):
    """one op of every kind (and their Controls) with real payloads; integer = small whole numbers (exact in every P)"""
    def vals(count):
        return rng.integers(-3, 4, size=count).astype(float) if integer else rng.standard_normal(count)

    pick = lambda k: [int(v) for v in rng.permutation(n)[:k]]  # noqa: E731
    ops = []
    for k in (1, 2, 3, 5):
        ops.append(MatrixOp.new_matrix(pick(k), vals(4**k)))
    idx = pick(3)
    ops.append(MatrixOp.new_sparse(idx, [[(int(c), float(v)) for c, v in zip(rng.integers(0, 8, size=2), vals(2))] for _ in range(8)]))
    ab = pick(4)
    ops.append(MatrixOp.new_swap(ab[:2], ab[2:]))
    c = pick(4)
    ops.append(MatrixOp.new_control(c[:2], c[2:], MatrixOp.new_matrix(c[2:], vals(16))))
    c = pick(5)
    ops.append(MatrixOp.new_control(c[:1], c[1:], MatrixOp.new_control(c[1:3], c[3:], MatrixOp.new_swap(c[3:4], c[4:]))))  # nested, uncollapsed
    c = pick(4)
    ops.append(MatrixOp.new_control(c[:1], c[1:], MatrixOp.new_sparse(c[1:], [[(int(j ^ 5), float(v))] for j, v in enumerate(vals(8))])))
    return ops


@pytest.mark.parametrize("dtype", REAL_TYPES)
def test_real_p_equals_the_real_part_of_the_complex_fold(dtype):
    """the complex oracle is pinned on the reference's vectors; with no imaginary part anywhere a complex fold IS the real fold
    ((a + 0i)(b + 0i) = ab - 0, 0; sums componentwise), so the real restatement must agree bit for bit — windows, accumulate and
    every op kind included"""
    n = 7
    rng = np.random.default_rng(5)
    integer = np.issubdtype(dtype, np.integer)
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    for op in random_real_ops(n, rng, integer):
        x = rng.integers(-4, 5, size=1 << n).astype(dtype) if integer else rng.standard_normal(1 << n).astype(dtype)
        y0 = rng.integers(-4, 5, size=1 << n).astype(dtype) if integer else rng.standard_normal(1 << n).astype(dtype)
        for (io, il, oo, ol) in ((0, 1 << n, 0, 1 << n), (16, 80, 8, 100), (0, 0, 3, 5), (100, 28, 0, 128)):
            for acc in (True, False):
                got = y0[:ol].copy()
                O.apply_op(n, op, np.ascontiguousarray(x[io:io + il]), got, io, oo, accumulate=acc)
                want = y0[:ol].astype(cdt)
                O.apply_op(n, op, np.ascontiguousarray(x[io:io + il]).astype(cdt), want, io, oo, accumulate=acc)
                assert np.all(want.imag == 0) and np.array_equal(got, want.real.astype(dtype)), (op, dtype, io, il, oo, ol, acc)
                if ol and not acc:  # apply_op_row (matrix_ops.rs:38-59) = one row of apply_op_overwrite
                    r = int(rng.integers(0, ol))
                    assert O.apply_op_row(n, op, np.ascontiguousarray(x[io:io + il]), r, io, oo) == got[r]


def test_integer_p_wraps():
    """i32 arithmetic is two's complement (Rust release builds): 2^30 * 4 = 0, 2^31 - 1 + 1 = -2^31"""
    x = np.array([1 << 30, (1 << 31) - 1], dtype=np.int32)
    out = np.array([0, 1], dtype=np.int32)
    O.apply_op(1, MatrixOp.new_matrix([0], [4, 0, 0, 1]), x, out)
    assert list(out) == [0, -(1 << 31)]


def test_real_payload_must_be_real():
    from rustqip_amd import CircuitError

    x = np.ones(2)
    with pytest.raises(CircuitError, match="imaginary"):
        O.apply_op(1, MatrixOp.new_matrix([0], [1j, 0, 0, 1]), x, np.zeros(2))
    with pytest.raises(CircuitError, match="fractional"):
        O.apply_op(1, MatrixOp.new_matrix([0], [0.5, 0, 0, 1]), x.astype(np.int64), np.zeros(2, dtype=np.int64))


def windows_accumulate(apply_op, n, op, x, dtype, parts_in=4, parts_out=2):
    """the reference's only provision for more than one device (matrix_ops.rs:96-97): `input` / `output` are WINDOWS of the 2^n
    vectors, a column outside the input window contributes zero — so accumulating apply_op over a partition of the input into
    windows, into each output window, rebuilds the whole product.  In integer arithmetic (a ring, wrapping included) exactly."""
    N = 1 << n
    out = np.zeros(N, dtype=dtype)
    for oo in range(0, N, N // parts_out):
        y = np.zeros(N // parts_out, dtype=dtype)
        for io in range(0, N, N // parts_in):
            apply_op(n, op, np.ascontiguousarray(x[io:io + N // parts_in]), y, io, oo)
        out[oo:oo + N // parts_out] = y
    return out


@pytest.mark.parametrize("dtype", (np.int64, np.int32))
def test_input_windows_accumulate_to_the_whole_vector(dtype):
    n = 8
    rng = np.random.default_rng(17)
    for op in random_real_ops(n, rng, True):
        x = rng.integers(-(1 << 20), 1 << 20, size=1 << n).astype(dtype)
        want = np.zeros(1 << n, dtype=dtype)
        O.apply_op(n, op, x, want)
        assert np.array_equal(windows_accumulate(O.apply_op, n, op, x, dtype), want), op
