"""Parity of the HIP path against the CPU oracle — needs a real MI355X (`-m gpu`).

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

pytestmark = pytest.mark.gpu

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


@pytest.fixture(scope="module")
def O():
    from oracle import qip_oracle

    return qip_oracle


def rand_state(n, seed, dtype=np.complex128):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return (v / np.linalg.norm(v)).astype(dtype)


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def hip_apply(n, op, x, **options):
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


# ---- the reference's own golden vectors, through the HIP path ---------------------------------------
def test_golden_kron_identities(O):
    def kron_helper(before, mat, after):
        eye = np.eye(2)
        for _ in range(before):
            mat = np.kron(eye, mat)
        for _ in range(after):
            mat = np.kron(mat, eye)
        return mat

    for data, qb in [([1, 0, 0, 1], 0), ([0, 1, 1, 0], 0), ([0, 1, 1, 0], 1), ([0, 1, 1, 0], 2), ([1, 2, 3, 4], 0)]:
        mat = q.make_op_matrix(3, MatrixOp.new_matrix([qb], data))
        assert np.array_equal(mat, kron_helper(qb, np.array(data, float).reshape(2, 2), 2 - qb).astype(complex))
    data = [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1]
    assert np.array_equal(q.make_op_matrix(4, MatrixOp.new_matrix([1, 2], data)),
                          kron_helper(1, np.array(data, float).reshape(4, 4), 1).astype(complex))
    data = list(range(16))
    comp = np.array(data, dtype=complex).reshape(4, 4)
    assert np.array_equal(q.make_op_matrix(2, MatrixOp.new_matrix([0, 1], data)), comp)
    assert not np.array_equal(q.make_op_matrix(2, MatrixOp.new_matrix([1, 0], data)), comp)


def test_golden_iterator_patterns_and_c64_cases():
    def cols(n, op):
        m = q.make_op_matrix(n, op)
        return [list(np.nonzero(m[r])[0]) for r in range(1 << n)]

    assert cols(1, MatrixOp.new_matrix([0], [0, 1, 1, 0])) == [[1], [0]]
    assert cols(1, MatrixOp.new_sparse([0], [[(1, 1)], [(0, 1)]])) == [[1], [0]]
    assert cols(2, MatrixOp.new_swap([0], [1])) == [[0], [2], [1], [3]]
    assert cols(2, MatrixOp.new_control([0], [1], MatrixOp.new_matrix([1], [0, 1, 1, 0]))) == [[0], [1], [3], [2]]
    inp = np.array([1, 0, 0, 0], dtype=np.complex128)
    out = np.zeros(4, dtype=np.complex128)
    q.apply_op(2, MatrixOp.new_matrix([0], [0, 1, 1, 0]), inp, out)
    assert np.array_equal(out, [0, 0, 1, 0])
    out = np.zeros(4, dtype=np.complex128)
    q.apply_op(2, MatrixOp.new_matrix([1], [0, 1, 1, 0]), inp, out)
    assert np.array_equal(out, [0, 1, 0, 0])
    inp = np.array([1, 0], dtype=np.complex128)
    out = np.zeros(2, dtype=np.complex128)
    q.apply_op(1, MatrixOp.new_matrix([0], [1, 0, 0, 1]), inp, out)
    assert np.array_equal(inp, out)


# ---- 1-qubit gates: every matrix class x every bit position x both low-bit variants ---------------
@pytest.mark.parametrize("name", sorted(GATES_1Q))
@pytest.mark.parametrize("n", [1, 2, 5, 11])
def test_single_qubit_all_targets(O, name, n):
    for target in range(n):
        op = q.make_matrix_op([target], GATES_1Q[name])
        x = rand_state(n, 100 + target)
        want = oracle_apply(O, n, op, x)
        for opts in ({}, {"lowbit_shuffle": 0}, {"force_generic": 1}):
            got = hip_apply(n, op, x, **opts)
            assert np.array_equal(got, want), (name, n, target, opts, np.max(np.abs(got - want)))


@pytest.mark.parametrize("nc", [1, 2, 3, 6])
def test_controlled_single_qubit(O, nc):
    n = 10
    rng = np.random.default_rng(nc)
    for trial in range(8):
        perm = [int(v) for v in rng.permutation(n)]
        ctrl, tgt = perm[:nc], perm[nc]
        for name in ("X", "H", "T", "Rz", "dense", "Y"):
            op = q.make_control_op(ctrl, q.make_matrix_op([tgt], GATES_1Q[name]))
            x = rand_state(n, trial)
            want = oracle_apply(O, n, op, x)
            for opts in ({}, {"lowbit_shuffle": 0}, {"force_generic": 1}):
                got = hip_apply(n, op, x, **opts)
                assert np.array_equal(got, want), (name, ctrl, tgt, opts)


def test_many_controls(O):
    n = 12
    op = q.make_control_op(list(range(n - 1)), q.make_matrix_op([n - 1], GATES_1Q["Z"]))
    check(O, n, op, exact=True)
    op = q.make_control_op(list(range(1, n)), q.make_matrix_op([0], GATES_1Q["H"]))
    check(O, n, op)
    # 15-control identity (qip/benches/state_bench.rs:172-186)
    op = q.make_control_op(list(range(n - 1)), q.make_matrix_op([n - 1], GATES_1Q["ident"]))
    check(O, n, op, exact=True)


def test_nested_control_uncollapsed(O):
    n = 6
    inner = MatrixOp.new_control([3], [5], MatrixOp.new_matrix([5], GATES_1Q["dense"]))
    op = MatrixOp.new_control([1], [3, 5], inner)
    check(O, n, op)
    flat = q.make_control_op([1, 3], q.make_matrix_op([5], GATES_1Q["dense"]))
    x = rand_state(n, 5)
    assert np.array_equal(hip_apply(n, op, x), hip_apply(n, flat, x))


# ---- Swap ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("h", [1, 2, 3])
def test_swap(O, h):
    n = 9
    rng = np.random.default_rng(h)
    for trial in range(6):
        perm = [int(v) for v in rng.permutation(n)]
        op = q.make_swap_op(perm[:h], perm[h:2 * h])
        check(O, n, op, seed=trial, exact=True)
        cop = q.make_control_op(perm[2 * h:2 * h + 2], op)
        check(O, n, cop, seed=trial, exact=True)


def test_swap_every_bit_pair_and_low_controls(O):
    """every (a, b) transposition at n = 9: lane-permutation form (both < 6), two-row cross-lane form
    (a < 6 <= b) and row-move form (both >= 6), bare and under controls that sit inside a 128-B line."""
    n = 9
    x = rand_state(n, 3)
    for qa in range(n):
        for qb in range(n):
            if qa == qb:
                continue
            op = q.make_swap_op([qa], [qb])
            assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x)), (qa, qb)
    for qa, qb, ctrl in ((8, 0, [7]), (8, 7, [6]), (0, 1, [8, 2]), (5, 2, [8, 7, 6]), (3, 4, [7, 0])):
        op = q.make_control_op(ctrl, q.make_swap_op([qa], [qb]))
        assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x)), (qa, qb, ctrl)


def test_swap_two_transpositions_per_sweep(O):
    """Swap(h >= 2) runs two transpositions per sweep (k_swap2): every combination of register-bit / lane-bit pairs
    (HH, HL, LL stages), with controls inside and outside a 128-B line, f64 / f32 (packed and unpacked view), against
    the oracle bit for bit and against the one-transposition-per-sweep path."""
    n = 12
    rng = np.random.default_rng(12)
    lo, hi = list(range(n - 6, n)), list(range(0, n - 6))  # qubits on bit positions 0..5 / 6..11
    shapes = {
        "HH,HH": lambda: (list(rng.permutation(hi)[:4]), []),
        "HH,HL": lambda: (list(rng.permutation(hi)[:3]), list(rng.permutation(lo)[:1])),
        "HH,LL": lambda: (list(rng.permutation(hi)[:2]), list(rng.permutation(lo)[:2])),
        "HL,HL": lambda: (list(rng.permutation(hi)[:2]), list(rng.permutation(lo)[:2])),
        "HL,LL": lambda: (list(rng.permutation(hi)[:1]), list(rng.permutation(lo)[:3])),
        "LL,LL": lambda: ([], list(rng.permutation(lo)[:4])),
    }
    for dtype in (np.complex128, np.complex64):
        x = rand_state(n, 5, dtype)
        for name, pick in shapes.items():
            for trial in range(4):
                H, L = pick()
                H, L = [int(v) for v in H], [int(v) for v in L]
                if name == "HH,HH":
                    a, b = [H[0], H[2]], [H[1], H[3]]
                elif name == "HH,HL":
                    a, b = [H[0], H[2]], [H[1], L[0]]
                elif name == "HH,LL":
                    a, b = [H[0], L[0]], [H[1], L[1]]
                elif name == "HL,HL":
                    a, b = [H[0], L[1]], [L[0], H[1]]
                elif name == "HL,LL":
                    a, b = [L[0], L[1]], [H[0], L[2]]
                else:
                    a, b = [L[0], L[2]], [L[1], L[3]]
                used = set(a + b)
                free = [t for t in range(n) if t not in used]
                for ctrl in ([], [free[0]], [t for t in free if t >= n - 3][:2], [free[-1], free[1]]):
                    op = q.make_swap_op(a, b)
                    if ctrl:
                        op = q.make_control_op(ctrl, op)
                    want = oracle_apply(O, n, op, x)
                    assert np.array_equal(hip_apply(n, op, x), want), (name, a, b, ctrl, dtype)
                    assert np.array_equal(hip_apply(n, op, x, swap_single=1), want), (name, a, b, ctrl)
                    if dtype == np.complex64:
                        assert np.array_equal(hip_apply(n, op, x, packed_f32=0), want), (name, a, b, ctrl)
    # h = 3 and 4: two sweeps
    x = rand_state(n, 6)
    for h in (3, 4, 5):
        for trial in range(6):
            perm = [int(v) for v in rng.permutation(n)]
            op = q.make_swap_op(perm[:h], perm[h:2 * h])
            assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x)), (h, perm)
    # small states fall back to one transposition per sweep
    for m in (4, 5, 6, 7, 8):
        xs = rand_state(m, m)
        op = q.make_swap_op([0, 1], [m - 1, m - 2])
        assert np.array_equal(hip_apply(m, op, xs), oracle_apply(O, m, op, xs)), m


def test_sparse_in_place_kernel(O):
    """SparseMatrix on k <= 5 qubits is applied in place (k_sparse_kq): rows in stored order, repeated columns, rows
    of very different lengths, targets on low bit positions, controls — bit-equal to the oracle, no second buffer."""
    n = 11
    rng = np.random.default_rng(7)
    x = rand_state(n, 7)
    for k in (1, 2, 3, 4, 5):
        for trial in range(5):
            perm = [int(v) for v in rng.permutation(n)]
            if trial == 0:
                perm = list(range(n - k, n)) + list(range(n - k))  # targets on the lowest bit positions
            rows = []
            for r in range(1 << k):
                cnt = int(rng.integers(1, (1 << k) + 3))
                cols = rng.integers(0, 1 << k, size=cnt)  # repeats allowed, arbitrary order
                rows.append([(int(c), complex(rng.standard_normal(), rng.standard_normal())) for c in cols])
            op = q.make_sparse_matrix_op(perm[:k], rows)
            for o in (op, q.make_control_op(perm[k:k + 1], op), q.make_control_op(perm[k:k + 2], op)):
                want = oracle_apply(O, n, o, x)
                with q.HipState(n) as st:
                    st.set_option("profile", 1)
                    st.upload(x)
                    p0 = st.device_ptr()
                    st.apply_op(o)
                    assert st.device_ptr() == p0  # in place: the buffers were not swapped
                    got = st.download()
                    assert "k_sparse_kq" in st.profile() or "k_sparse_tile" in st.profile(), st.profile()  # (r4: k = 4, 5 with narrow rows may take the tile form)
                assert np.array_equal(got, want), (k, trial, repr(o))
        xf = rand_state(n, 8, np.complex64)
        rows = [[(int(c), complex(rng.standard_normal(), rng.standard_normal())) for c in rng.integers(0, 1 << k, size=2)] for _ in range(1 << k)]
        op = q.make_sparse_matrix_op([int(v) for v in rng.permutation(n)[:k]], rows)
        assert np.array_equal(hip_apply(n, op, xf), oracle_apply(O, n, op, xf)), k
    # k = 6 stays on the literal kernel
    rows = [[((r + 1) % 64, 1j)] for r in range(64)]
    check(O, 8, q.make_sparse_matrix_op([0, 7, 2, 5, 4, 3], rows))
    # a program with a sparse op is a graph now (nothing swaps buffers)
    ops = [q.make_matrix_op([0], circuits.H), q.make_sparse_matrix_op([1, 9], [[(1, 1j)], [(0, 1.0)], [(3, 1.0)], [(2, -1.0)]])]
    with q.HipState(n) as st:
        st.upload(x)
        prog = st.compile_program(ops)
        prog.run()
        assert prog.is_graph
        assert np.array_equal(st.download(), O.apply_ops_in_place(n, ops, x.copy()))
        prog.close()


def test_program_survives_arena_regrowth(O):
    """a captured graph bakes in the device-arena address; an eager op that needs a larger payload frees
    and regrows the arena (here: two dense k = 7 ops, which also restore `cur`).  The replay must re-record."""
    n = 10
    rng = np.random.default_rng(3)
    x = rand_state(n, 9)
    u2 = rand_unitary(2, rng)
    ops = [q.make_matrix_op([0, 5], u2.ravel()), q.make_matrix_op([3], circuits.H),
           q.make_matrix_op([1, 2, 9], np.diag(np.exp(1j * rng.uniform(0, 6, 8))).ravel())]
    u7 = rand_unitary(7, rng)
    big = q.make_matrix_op([0, 1, 2, 3, 4, 5, 6], u7.ravel())
    with q.HipState(n) as st:
        st.upload(x)
        prog = st.compile_program(ops)
        prog.run()
        assert prog.is_graph
        st.apply_op(big)
        st.apply_op(big)
        prog.run()
        got = st.download()
        prog.close()
    want = O.apply_ops_in_place(n, ops + [big, big] + ops, x.copy())
    assert np.max(np.abs(got - want)) <= TOL64


def test_measure_probs_many_outcomes(O):
    """5 <= k <= 16 measured qubits, any mix of low / high bit positions and any outcome-bit order
    (k_measure_probs_grid: outcomes on the grid + per-lane fold, no atomics)."""
    n = 18
    rng = np.random.default_rng(18)
    x = rand_state(n, 18)
    x[::7] = 0  # zero amplitudes are skipped by the reference (measurement_ops.rs:98-99)
    x /= np.linalg.norm(x)
    with q.HipState(n) as st:
        st.upload(x)
        cases = [list(range(12)), list(range(n - 8, n)), list(range(n - 9, n - 1))[::-1], list(range(n))[:16]]
        for k in (5, 6, 8, 9, 11, 13, 16):
            cases.append([int(v) for v in rng.permutation(n)[:k]])
        for idx in cases:
            got = st.measure_probs(idx)
            want = O.measure_probs(n, idx, x)
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 1e-13, idx
            assert abs(got.sum() - 1) < 1e-12
    xf = rand_state(12, 2, np.complex64)
    with q.HipState(12, np.complex64) as st:
        st.upload(xf)
        for idx in ([0, 11, 5, 6, 7, 1], list(range(12))):
            assert np.max(np.abs(st.measure_probs(idx) - O.measure_probs(12, idx, xf))) <= 1e-5
    # Complex<f32> is read as 16-byte elements of two amplitudes; r4: also when index bit 0 (qubit n-1) is measured — the halves of
    # an element go to two outcomes.  Bit 0 as the first / last / a middle outcome bit, with and without other row positions, k = 5..16
    xf = rand_state(n, 3, np.complex64)
    xf[::5] = 0
    with q.HipState(n, np.complex64) as st:
        st.upload(xf)
        cases = [[n - 1, 0, 3, 9, 12], [0, 3, 9, 12, n - 1], [4, n - 1, n - 2, 7, n - 5, 1, 10], list(range(n - 12, n)), list(range(n - 1, n - 13, -1)),
                 [0, n - 1, 3, n - 4, 7, n - 9, 11, n - 13, 15, 13, n - 2, 1], list(range(2, n))]
        for k in (5, 8, 11, 14):
            c = [int(v) for v in rng.permutation(n - 1)[:k - 1]]
            c.insert(int(rng.integers(0, k)), n - 1)
            cases.append(c)
        for idx in cases:
            got = st.measure_probs(idx)
            want = O.measure_probs(n, idx, xf)
            assert np.max(np.abs(got - want)) <= 2e-6 * max(1.0, float(np.max(want)) * (1 << len(idx)) / 64), idx
            assert abs(got.sum() - float(np.sum(np.abs(xf.astype(np.complex128)) ** 2))) < 1e-5


@pytest.fixture
def line_bits(request):
    q.set_global_option("line_bits", request.param)
    yield request.param
    q.set_global_option("line_bits", 3)


@pytest.mark.parametrize("line_bits", [3, 2, 1, 0], indirect=True)
def test_selectors_inside_a_cache_line(O, line_bits):
    """controls / phase bits at bit positions below `line_bits` become lane predicates (full-line sweeps); at or above
    it they are removed from the grid (only the matching sub-space is swept).  Every threshold is bit-equal."""
    n = 10
    x = rand_state(n, 4)
    low_q = [n - 1, n - 2, n - 3]  # qubits at bit positions 0, 1, 2
    for ctrl in ([low_q[0]], [low_q[1]], [low_q[2]], low_q[:2], low_q, [low_q[0], 1], [low_q[2], 0, 4]):
        for tgt in (0, 5, 3):
            for name in ("X", "H", "Rz", "T", "Z", "dense"):
                op = q.make_control_op(ctrl, q.make_matrix_op([tgt], GATES_1Q[name]))
                want = oracle_apply(O, n, op, x)
                for opts in ({}, {"lowbit_shuffle": 0}):
                    assert np.array_equal(hip_apply(n, op, x, **opts), want), (ctrl, tgt, name, opts)
        # low target with low control (cross-lane kernel with a predicate)
        free_low = [t for t in low_q if t not in ctrl]
        if free_low:
            op = q.make_control_op(ctrl, q.make_matrix_op([free_low[0]], GATES_1Q["H"]))
            assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x))
    for tq in low_q + [n - 4, 0]:
        for name in ("T", "Z", "S", "Rz"):
            op = q.make_matrix_op([tq], GATES_1Q[name])
            assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x)), (tq, name)
    rng = np.random.default_rng(1)
    d = np.exp(1j * rng.uniform(0, 6, 8))
    d[3] = 1.0
    for idx in ([n - 1, n - 2, 0], [0, n - 1, 4], [n - 3, n - 2, n - 1]):
        op = q.make_matrix_op(idx, np.diag(d).ravel())
        assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x)), idx
        cop = q.make_control_op([5 if 5 not in idx else 6], op)
        assert np.array_equal(hip_apply(n, cop, x), oracle_apply(O, n, cop, x)), idx


# ---- dense k-qubit --------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [2, 3, 4, 5, 6])
def test_dense_k_qubit(O, k):
    """k = 2: register kernel (bit-equal).  k = 3..5 on f64: matrix-core kernel (fma chains, so the
    stated 1e-12 bar applies, not bit equality); with option mfma = 0 the VALU register kernel
    (k <= 4) / the literal kernel (k >= 5) must again be bit-equal."""
    n = 10
    rng = np.random.default_rng(10 + k)
    for trial in range(6):
        perm = [int(v) for v in rng.permutation(n)]
        if trial == 4:
            perm = list(range(n))[::-1]  # targets on the lowest bit positions
        if trial == 5:
            perm = list(range(n))        # targets on the highest bit positions
        u = rand_unitary(k, rng)
        op = q.make_matrix_op(perm[:k], u.ravel())
        x = rand_state(n, trial)
        want = oracle_apply(O, n, op, x)
        got = hip_apply(n, op, x)
        if k == 2:
            assert np.array_equal(got, want)
        else:
            assert np.max(np.abs(got - want)) <= TOL64, (k, perm[:k])
        assert np.array_equal(hip_apply(n, op, x, mfma=0), want)
        assert np.array_equal(hip_apply(n, op, x, force_generic=1), want)
        if k <= 5 and n - k >= 6:
            cop = q.make_control_op(perm[k:k + 2], op)
            want = oracle_apply(O, n, cop, x)
            assert np.max(np.abs(hip_apply(n, cop, x) - want)) <= TOL64
            assert np.array_equal(hip_apply(n, cop, x, mfma=0), want)


@pytest.mark.parametrize("k", [6, 7, 8, 9, 10])
def test_dense_big_k_streamed_matrix_core_kernel(O, k):
    """dense k = 6..8 on f64: k_gate_big_mfma (A operand streamed through LDS, X in registers, in place); r4: k = 9, 10:
    k_gate_huge_mfma (X in LDS, A streamed from L2 in pairs of K-steps; 8 groups per item for Complex<f64> at k = 10) — targets on
    low / high / mixed bit positions, with controls, n from the smallest size the kernel accepts (k + 4) upwards;
    fma chains, so the 1e-12 bar; 0/1 permutation matrices stay exact; mfma = 0 still takes the literal kernel."""
    rng = np.random.default_rng(100 + k)
    u = rand_unitary(k, rng)
    for n in (k + 4, k + 5, k + 7):
        x = rand_state(n, n)
        picks = [list(range(k)), list(range(n - k, n)), [int(v) for v in rng.permutation(n)[:k]], [int(v) for v in rng.permutation(n)[:k]]]
        for idx in picks:
            op = q.make_matrix_op(idx, u.ravel())
            want = oracle_apply(O, n, op, x)
            with q.HipState(n) as st:
                st.set_option("profile", 1)
                st.upload(x)
                p0 = st.device_ptr()
                st.apply_op(op)
                got = st.download()
                assert "k_gate_big_mfma" in st.profile() and st.device_ptr() == p0, st.profile()
            assert np.max(np.abs(got - want)) <= TOL64, (k, n, idx)
        if n >= k + 5:
            free = [t for t in range(n) if t not in picks[2]]
            cop = q.make_control_op(free[:1], q.make_matrix_op(picks[2], u.ravel()))
            assert np.max(np.abs(hip_apply(n, cop, x) - oracle_apply(O, n, cop, x))) <= TOL64
    n = k + 5
    x = rand_state(n, 3)
    perm = rng.permutation(1 << k)
    pm = np.zeros((1 << k, 1 << k))
    pm[np.arange(1 << k), perm] = 1
    op = q.make_matrix_op([int(v) for v in rng.permutation(n)[:k]], pm.ravel())
    assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x))
    op = q.make_matrix_op(list(range(k)), u.ravel())
    assert np.array_equal(hip_apply(n, op, x, mfma=0), oracle_apply(O, n, op, x))  # literal kernel: bit-equal
    # the f32 form of the same kernel (v_mfma_f32_16x16x4_f32)
    xf = rand_state(n, 4, np.complex64)
    for idx in (list(range(n - k, n)), [int(v) for v in rng.permutation(n)[:k]]):
        opf = q.make_matrix_op(idx, u.astype(np.complex64).ravel())
        with q.HipState(n, np.complex64) as st:
            st.set_option("profile", 1)
            st.upload(xf)
            st.apply_op(opf)
            got = st.download()
            assert "k_gate_big_mfma" in st.profile()
        assert np.max(np.abs(got - oracle_apply(O, n, opf, xf))) <= TOL32, (k, idx)


def test_dense_k_qubit_f32_matrix_cores(O):
    """f32 states: dense k = 3..5 on v_mfma_f32_16x16x4_f32 (exact f32 fma chains; the C/D layout differs from the f64
    form and the host arranges the A rows for it) — 1e-5 bar vs the f32 oracle, 0/1 permutation matrices exact, k = 5 no
    longer on the literal kernel."""
    n = 11
    rng = np.random.default_rng(55)
    x = rand_state(n, 5, np.complex64)
    for k in (3, 4, 5):
        u = rand_unitary(k, rng).astype(np.complex64)
        for idx in (list(range(n - k, n)), list(range(k)), [int(v) for v in rng.permutation(n)[:k]], [n - 1, n - 2] + [int(v) for v in rng.permutation(n - 2)[:k - 2]]):
            op = q.make_matrix_op(idx, u.ravel())
            want = oracle_apply(O, n, op, x)
            with q.HipState(n, np.complex64) as st:
                st.set_option("profile", 1)
                st.set_option("mfma", 2)  # force the matrix-core form also where the register form would be chosen
                st.upload(x)
                st.apply_op(op)
                got = st.download()
                assert "k_gate_kq_mfma" in st.profile(), st.profile()
            assert np.max(np.abs(got - want)) <= TOL32, (k, idx)
            if k == 5:
                with q.HipState(n, np.complex64) as st:
                    st.set_option("profile", 1)
                    st.upload(x)
                    st.apply_op(op)
                    assert "k_gate_kq_mfma" in st.profile()  # the default path for k = 5
            cop = q.make_control_op([t for t in range(n) if t not in idx][:1], op)
            assert np.max(np.abs(hip_apply(n, cop, x) - oracle_apply(O, n, cop, x))) <= TOL32
        perm = rng.permutation(1 << k)
        pm = np.zeros((1 << k, 1 << k))
        pm[np.arange(1 << k), perm] = 1
        op = q.make_matrix_op([int(v) for v in rng.permutation(n)[:k]], pm.ravel())
        assert np.array_equal(hip_apply(n, op, x, mfma=2), oracle_apply(O, n, op, x))


def test_dense_permutations_stay_exact_on_matrix_cores(O):
    """0/1 permutation matrices through the MFMA path: fma(1, x, 0) is exact, so IEEE `==` holds."""
    n = 9
    toffoli = np.eye(8)
    toffoli[6:, 6:] = [[0, 1], [1, 0]]
    fredkin = np.eye(8)
    fredkin[[5, 6]] = fredkin[[6, 5]]
    cyc = np.roll(np.eye(16), 3, axis=0)
    for mat, idxs in ((toffoli, ([0, 4, 8], [8, 7, 6], [2, 0, 1])), (fredkin, ([1, 2, 3], [8, 0, 4])),
                      (cyc, ([0, 1, 2, 3], [8, 6, 4, 2]))):
        for idx in idxs:
            check(O, n, q.make_matrix_op(idx, mat.ravel()), exact=True)


def test_matrix_core_kernel_sizes(O):
    """k = 3 on the smallest state the MFMA path accepts (n = k + 4) and one below it (fallback)."""
    rng = np.random.default_rng(77)
    for n in (6, 7, 8, 12):
        u = rand_unitary(3, rng)
        op = q.make_matrix_op([n - 1, 0, n // 2], u.ravel())
        x = rand_state(n, n)
        assert np.max(np.abs(hip_apply(n, op, x) - oracle_apply(O, n, op, x))) <= TOL64
    u5 = rand_unitary(5, rng)
    for n in (8, 9, 13):
        op = q.make_matrix_op([n - 1, 0, 3, 2, n - 2], u5.ravel())
        x = rand_state(n, n)
        assert np.max(np.abs(hip_apply(n, op, x) - oracle_apply(O, n, op, x))) <= TOL64


def test_two_qubit_permutation_matrix_exact(O):
    data = [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1]  # matrix_ops.rs:323-334
    for idx in ([1, 2], [6, 0], [7, 8]):
        check(O, 9, q.make_matrix_op(idx, data), exact=True)


def test_dense_8_qubit_reference_bench_shape(O):
    # qip/benches/state_bench.rs:118-139: dense 8-qubit matrix on an 8-qubit state
    rng = np.random.default_rng(8)
    u = rand_unitary(8, rng)
    check(O, 8, q.make_matrix_op(list(range(8)), u.ravel()), paths=("fast",))


# ---- diagonal -------------------------------------------------------------------------------------
def test_diagonal_gates(O):
    n = 9
    rng = np.random.default_rng(3)
    for k in (2, 3, 5):
        for trial in range(4):
            perm = [int(v) for v in rng.permutation(n)]
            d = np.exp(1j * rng.uniform(0, 2 * np.pi, 1 << k))
            if trial % 2:
                d[rng.integers(0, 1 << k)] = 1.0  # exercise the skip-unit-entries path
            op = q.make_matrix_op(perm[:k], np.diag(d).ravel())
            check(O, n, op, seed=trial)
            check(O, n, q.make_control_op(perm[k:k + 1], op), seed=trial)
    # controlled-phase ladder element of the QFT
    op = q.make_control_op([7], q.make_matrix_op([2], [1, 0, 0, cmath.rect(1, math.pi / 32)]))
    check(O, n, op)
    # diag(phase, 1): the non-unit entry sits on |0>
    check(O, n, q.make_matrix_op([4], [cmath.rect(1, 0.2), 0, 0, 1]))


# ---- SparseMatrix ---------------------------------------------------------------------------------
def test_sparse(O):
    n = 8
    rng = np.random.default_rng(4)
    for k in (1, 2, 4):
        for trial in range(4):
            perm = [int(v) for v in rng.permutation(n)]
            rows = []
            for r in range(1 << k):
                cnt = int(rng.integers(1, min(4, 1 << k) + 1))
                cols = rng.choice(1 << k, size=cnt, replace=False)  # stored order is arbitrary
                rows.append([(int(c), complex(rng.standard_normal(), rng.standard_normal())) for c in cols])
            op = q.make_sparse_matrix_op(perm[:k], rows)
            check(O, n, op, seed=trial)
            check(O, n, q.make_control_op(perm[k:k + 2], op), seed=trial)
    # 16-qubit sparse identity of the reference bench, shrunk (state_bench.rs:380-393)
    ident = q.make_sparse_matrix_op(list(range(8)), [[(r, 1.0)] for r in range(256)])
    check(O, 8, ident, exact=True)


def test_repeated_indices_follow_reference_literally(O):
    # the reference accepts repeated indices; the literal gather kernel reproduces its index math
    for op in (MatrixOp.new_matrix([1, 1], np.arange(16).astype(complex)),
               MatrixOp.new_swap([2], [2]),
               MatrixOp.new_control([0], [0], MatrixOp.new_matrix([0], [0, 1, 1, 0]))):
        check(O, 4, op, paths=("fast",))


# ---- host-pointer twin: windows, offsets, accumulate --------------------------------------------------
def test_host_twin_accumulate_and_overwrite(O):
    n = 7
    rng = np.random.default_rng(5)
    x = rand_state(n, 1)
    for op in (q.make_matrix_op([3], GATES_1Q["H"]), q.make_control_op([0], q.make_matrix_op([6], GATES_1Q["X"])),
               q.make_swap_op([1], [5]), q.make_matrix_op([2, 4], rand_unitary(2, rng).ravel())):
        base = rand_state(n, 2)
        for generic in (0, 1):
            q.set_global_option("force_generic", generic)
            try:
                got = base.copy()
                q.apply_op(n, op, x, got)
                want = base.copy()
                O.apply_op(n, op, x, want)
                assert np.array_equal(got, want)
                got = base.copy()
                q.apply_op_overwrite(n, op, x, got)
                want = base.copy()
                O.apply_op_overwrite(n, op, x, want)
                assert np.array_equal(got, want)
            finally:
                q.set_global_option("force_generic", 0)


def test_host_twins_of_apply_op_row_and_windowed_measurement(O):
    """apply_op_row (matrix_ops.rs:38-59) and measure_prob / measure_probs with an input_offset window
    (measurement_ops.rs:44-58,115-127) through the C ABI's host twins, against the oracle's restatements."""
    from rustqip_amd.state import apply_op_row, measure_prob, measure_probs

    n = 8
    rng = np.random.default_rng(8)
    x = rand_state(n, 8)
    ops = [q.make_matrix_op([3], GATES_1Q["dense"]), q.make_control_op([0, 7], q.make_matrix_op([2], GATES_1Q["H"])),
           q.make_swap_op([1], [6]), q.make_matrix_op([5, 0], rand_unitary(2, rng).ravel()),
           q.make_sparse_matrix_op([2, 4], [[(1, 0.5j)], [(0, 2.0), (3, 1.0)], [(3, 1.0)], [(2, -1.0)]])]
    for op in ops:
        for row in (0, 1, 77, 255):
            assert apply_op_row(n, op, x, row) == O.apply_op_row(n, op, x, row), (repr(op), row)
        # windows: input = amplitudes [64, 192), rows addressed relative to output_offset 100
        win = np.ascontiguousarray(x[64:192])
        for row in (0, 5, 60):
            assert apply_op_row(n, op, win, row, 64, 100) == O.apply_op_row(n, op, win, row, 64, 100), (repr(op), row)
    for idx in ([0], [7, 0], [3, 4, 5], [6, 1, 0, 2]):
        for off, ln in ((0, 256), (64, 128), (100, 37), (255, 1), (17, 0)):
            win = np.ascontiguousarray(x[off:off + ln])
            got = measure_probs(n, idx, win, off)
            want = O.measure_probs(n, idx, win, off) if ln else np.zeros(1 << len(idx))
            assert np.max(np.abs(got - want)) <= 1e-14, (idx, off, ln)
            m = int(rng.integers(0, 1 << len(idx)))
            assert abs(measure_prob(n, m, idx, win, off) - want[m]) <= 1e-14
    # the shard identity of the reference's windows: the shards' windowed probabilities add up to the whole
    parts = [measure_probs(n, [0, 5], np.ascontiguousarray(x[r * 64:(r + 1) * 64]), r * 64) for r in range(4)]
    assert np.max(np.abs(sum(parts) - O.measure_probs(n, [0, 5], x))) <= 1e-14
    xf = rand_state(n, 9, np.complex64)
    assert np.max(np.abs(measure_probs(n, [1, 2], xf[32:96].copy(), 32) - O.measure_probs(n, [1, 2], xf[32:96].copy(), 32))) <= 1e-6


def test_host_twin_shard_window_identity(O):
    """shard r of the result = sum over windows w of apply_op(in_w -> out_r, w*S, r*S)."""
    n, shards = 8, 4
    S = (1 << n) // shards
    rng = np.random.default_rng(6)
    x = rand_state(n, 3)
    for op in (q.make_matrix_op([0], GATES_1Q["H"]), q.make_matrix_op([1, 6], rand_unitary(2, rng).ravel()),
               q.make_control_op([1], q.make_matrix_op([0], GATES_1Q["X"])), q.make_swap_op([0], [7])):
        full = oracle_apply(O, n, op, x)
        for r in range(shards):
            out = np.zeros(S, dtype=np.complex128)
            ref = np.zeros(S, dtype=np.complex128)
            for w in range(shards):
                q.apply_op(n, op, x[w * S:(w + 1) * S].copy(), out, w * S, r * S)
                O.apply_op(n, op, x[w * S:(w + 1) * S].copy(), ref, w * S, r * S)
            assert np.array_equal(out, ref)
            assert np.max(np.abs(out - full[r * S:(r + 1) * S])) < 1e-14


def test_host_twin_ragged_windows(O):
    n = 6
    rng = np.random.default_rng(7)
    op = q.make_matrix_op([2, 5], rand_unitary(2, rng).ravel())
    for in_len, out_len, in_off, out_off in [(10, 7, 3, 40), (64, 1, 0, 63), (1, 64, 17, 0), (0, 5, 0, 2), (5, 0, 1, 1)]:
        x = rand_state(n, 9)[:in_len].copy()
        got = np.full(out_len, 2.0 + 1j, dtype=np.complex128)
        want = got.copy()
        q.apply_op(n, op, x, got, in_off, out_off)
        O.apply_op(n, op, x, want, in_off, out_off)
        assert np.array_equal(got, want)


def test_invalid_descriptors_are_errors_not_crashes():
    x = np.zeros(4, dtype=np.complex128)
    out = np.zeros(4, dtype=np.complex128)
    with pytest.raises(q.CircuitError):
        q.apply_op(2, MatrixOp.new_matrix([2], [0, 1, 1, 0]), x, out)  # index >= n
    with pytest.raises(q.CircuitError):
        q.apply_op(2, MatrixOp.new_matrix([0], [0, 1, 1]), x, out)  # wrong data length
    with pytest.raises(q.CircuitError):
        q.apply_op(2, MatrixOp.new_sparse([0], [[(0, 1)], [(5, 1)]]), x, out)  # column out of range
    with pytest.raises(q.CircuitError):
        q.apply_op(2, MatrixOp.new_matrix([0], [0, 1, 1, 0]), x, x)  # aliasing


# ---- f32 ----------------------------------------------------------------------------------------------
def test_complex64_path(O):
    n = 9
    rng = np.random.default_rng(8)
    ops = [q.make_matrix_op([t], GATES_1Q[g]) for t in (0, 4, 8) for g in ("H", "X", "Rz", "T")]
    ops += [q.make_control_op([2], q.make_matrix_op([7], GATES_1Q["X"])), q.make_swap_op([0], [8]),
            q.make_matrix_op([1, 5], rand_unitary(2, rng).ravel()), q.make_matrix_op([1, 5, 6], rand_unitary(3, rng).ravel())]
    for op in ops:
        x = rand_state(n, 11, np.complex64)
        want = oracle_apply(O, n, op, x)
        for opts in ({}, {"force_generic": 1}):
            got = hip_apply(n, op, x, **opts)
            assert got.dtype == np.complex64
            assert np.max(np.abs(got - want)) <= TOL32


def test_complex64_packed_view(O):
    """f32 states are swept as 2^(n-1) 16-B elements of two amplitudes whenever index bit 0 is not a
    selector; bit 0 as a 1-qubit target is handled inside the element.  Same arithmetic as the unpacked
    8-B path, so the two must be bit-identical, and both match the f32 oracle."""
    n = 9
    rng = np.random.default_rng(12)
    x = rand_state(n, 13, np.complex64)
    ops = []
    for t in range(n):
        for g in ("H", "X", "Rz", "T", "Z", "dense", "upper"):
            ops.append(q.make_matrix_op([t], GATES_1Q[g]))
    for c, t in ((0, 8), (8, 0), (7, 8), (8, 7), (3, 6), (6, 3), (7, 1)):
        for g in ("X", "H", "Rz", "T"):
            ops.append(q.make_control_op([c], q.make_matrix_op([t], GATES_1Q[g])))
    ops.append(q.make_control_op([0, 7, 2], q.make_matrix_op([8], GATES_1Q["X"])))
    for a, b in ((0, 8), (8, 7), (7, 6), (2, 5), (1, 7), (0, 1)):
        ops.append(q.make_swap_op([a], [b]))
        ops.append(q.make_control_op([4], q.make_swap_op([a], [b])))
    d = np.exp(1j * rng.uniform(0, 6, 4))
    for idx in ([0, 1], [7, 8], [3, 8], [7, 2]):
        ops.append(q.make_matrix_op(idx, np.diag(d).ravel()))
    for op in ops:
        want = oracle_apply(O, n, op, x)
        packed = hip_apply(n, op, x)
        plain = hip_apply(n, op, x, packed_f32=0)
        assert packed.dtype == np.complex64
        assert np.array_equal(packed, plain), repr(op)
        assert np.max(np.abs(packed - want)) <= TOL32, repr(op)
    # a whole circuit in f32, n large enough for the unguarded kernel shapes
    n = 16
    circ = circuits.h_layer(n) + circuits.c2_random_circuit(n, 128, seed=5) + circuits.c3_qft(n)[:60]
    x = circuits.random_state(n, 3, np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.upload(x)
        st.apply_ops(circ)
        got = st.download()
    with q.HipState(n, np.complex64) as st:
        st.set_option("packed_f32", 0)
        st.upload(x)
        st.apply_ops(circ)
        plain = st.download()
    assert np.array_equal(got, plain)
    want = O.apply_ops_in_place(n, circ, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-4


# ---- whole circuits (BASELINE configs at reduced n) -------------------------------------------------------
@pytest.mark.parametrize("name,n", [("c2", 16), ("c3", 12), ("c4", 14), ("c5", 10), ("c5k3", 10)])
def test_config_circuits_reduced_n(O, name, n):
    ops = {
        "c2": lambda: circuits.h_layer(n) + circuits.c2_random_circuit(n, 256, seed=28),
        "c3": lambda: circuits.c3_qft(n),
        "c4": lambda: circuits.h_layer(n) + circuits.c4_clifford_t(n, 256, seed=32),
        "c5": lambda: circuits.h_layer(n) + circuits.c5_grover_iteration(n),
        "c5k3": lambda: circuits.h_layer(n) + circuits.c5_grover_iteration(n, dense_k3=True),
    }[name]()
    x = circuits.random_state(n, seed=n) if name == "c3" else None
    with q.HipState(n) as st:
        if x is None:
            st.init_basis(0)
            x = np.zeros(1 << n, dtype=np.complex128)
            x[0] = 1
        else:
            st.upload(x)
        st.apply_ops(ops)
        got = st.download()
        norm = st.norm_sqr()
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= TOL64
    assert abs(norm - 1.0) <= TOL64
    if name in ("c5", "c5k3"):
        # one Grover iteration amplifies the marked item |0..0>: sin^2(3*theta), sin(theta) = 2^(-n/2)
        theta = math.asin(2 ** (-n / 2))
        assert abs(abs(got[0]) ** 2 - math.sin(3 * theta) ** 2) < 1e-10


@pytest.mark.parametrize("K", [2, 3, 4, 5])
def test_gate_fusion_matches_gate_by_gate(O, K):
    """option fuse = K: consecutive gates merged into dense <= K-qubit gates, one sweep each.
    Same state as the gate-by-gate oracle to the 1e-12 bar (matrix products round differently)."""
    n = 12
    rng = np.random.default_rng(K)
    mixed = []
    for _ in range(40):
        perm = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 7))
        if kind == 0:
            mixed.append(q.make_swap_op([perm[0]], [perm[1]]))
        elif kind == 1:
            mixed.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], GATES_1Q["dense"])))
        elif kind == 2:
            mixed.append(q.make_sparse_matrix_op(perm[:2], [[(1, 0.5j)], [(0, 2.0)], [(3, 1.0)], [(2, -1.0), (3, 0.25)]]))
        elif kind == 3:
            mixed.append(q.make_control_op(perm[:7], q.make_matrix_op([perm[7]], GATES_1Q["Z"])))  # not fusable
        elif kind == 4:
            mixed.append(q.make_matrix_op(perm[:3], rand_unitary(3, rng).ravel()))
        else:
            mixed.append(q.make_matrix_op([perm[0]], GATES_1Q[["H", "T", "Rz", "X"][int(rng.integers(0, 4))]]))
    for name, ops in (("c2", circuits.h_layer(n) + circuits.c2_random_circuit(n, 200, seed=28)),
                      ("qft", circuits.c3_qft(n)),
                      ("c4", circuits.c4_clifford_t(n, 200, seed=32)),
                      ("grover", circuits.h_layer(n) + circuits.c5_grover_iteration(n)),
                      ("mixed", mixed)):
        x = circuits.random_state(n, seed=K)
        with q.HipState(n) as st:
            st.set_option("fuse", K)
            st.set_option("profile", 1)
            st.upload(x)
            st.apply_ops(ops)
            got = st.download()
            sweeps = sum(v["launches"] for v in st.profile().values())
        want = O.apply_ops_in_place(n, ops, x.copy())
        scale = max(1.0, float(np.max(np.abs(want))))
        assert np.max(np.abs(got - want)) <= TOL64 * scale * 10, (name, K)
        if name in ("c2", "c4"):
            assert sweeps < len(ops) / 1.5, (name, K, sweeps, len(ops))  # fusion really merged gates


@pytest.mark.parametrize("n", [12, 13, 16])
def test_lds_tile_multi_gate_sweeps(O, n):
    """option tile: whole segments of gates applied in one LDS-resident sweep.  tile = 1 keeps the circuit's
    gate order and must be BIT-IDENTICAL to the gate-by-gate path; tile = 2 (commuting reorder) meets 1e-12."""
    rng = np.random.default_rng(n)
    mixed = []
    for _ in range(120):
        perm = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 9))
        if kind == 0:
            mixed.append(q.make_swap_op([perm[0]], [perm[1]]))
        elif kind == 1:
            mixed.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], GATES_1Q["dense"])))
        elif kind == 2:
            mixed.append(q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel()))  # not tileable
        elif kind == 3:
            mixed.append(q.make_control_op([perm[0]], q.make_swap_op([perm[1]], [perm[2]])))
        elif kind == 4:
            mixed.append(q.make_control_op(perm[:1], q.make_matrix_op([perm[1]], [1, 0, 0, cmath.rect(1, 0.7)])))
        else:
            mixed.append(q.make_matrix_op([perm[0]], GATES_1Q[["H", "T", "Rz", "X", "Y", "upper", "S"][int(rng.integers(0, 7))]]))
    for name, ops in (("c2", circuits.h_layer(n) + circuits.c2_random_circuit(n, 200, seed=28)),
                      ("qft", circuits.c3_qft(n)),
                      ("c4", circuits.c4_clifford_t(n, 200, seed=32)),
                      ("grover", circuits.h_layer(n) + circuits.c5_grover_iteration(n)),
                      ("mixed", mixed)):
        x = circuits.random_state(n, seed=n)
        with q.HipState(n) as st:
            st.upload(x)
            st.apply_ops(ops)
            eager = st.download()
        with q.HipState(n) as st:  # one LDS round trip per gate (the simpler kernel) is bit-identical too
            st.set_option("tile", 1)
            st.set_option("tile_passes", 0)
            st.upload(x)
            st.apply_ops(ops)
            assert np.array_equal(st.download(), eager), (name, n)
        sweeps = {}
        for mode in (1, 2):
            with q.HipState(n) as st:
                st.set_option("tile", mode)
                st.set_option("profile", 1)
                st.upload(x)
                st.apply_ops(ops)
                got = st.download()
                sweeps[mode] = sum(v["launches"] for v in st.profile().values())
            if mode == 1:
                assert np.array_equal(got, eager), (name, n)
            else:
                assert np.max(np.abs(got - eager)) <= TOL64 * max(1.0, float(np.max(np.abs(eager)))), (name, n)
        if name in ("c2", "c4", "qft"):
            assert sweeps[1] < len(ops) / 2 and sweeps[2] <= sweeps[1], (name, sweeps, len(ops))
    want = O.apply_ops_in_place(n, mixed, x.copy())
    with q.HipState(n) as st:
        st.set_option("tile", 1)
        st.upload(x)
        st.apply_ops(mixed)
        assert np.array_equal(st.download(), want)  # and bit-equal to the oracle itself


def test_tile_segments_compiled_at_run_time_are_bit_identical(O):
    """option tile_jit: each tile segment runs as a kernel compiled for that very segment (hiprtc, cached by source).
    Same helpers, same order of operations => IEEE-identical to the interpreter kernel, for f64 and f32, eagerly and as a
    captured program; a segment met again is not compiled again."""
    import ctypes as C

    from rustqip_amd import _ffi

    def jit_count():
        k, ms = C.c_uint64(), C.c_double()
        assert _ffi.lib.qip_hip_jit_stats(C.byref(k), C.byref(ms)) == 0
        return int(k.value), ms.value

    rng = np.random.default_rng(77)
    for n, dtype in ((13, np.complex128), (16, np.complex128), (14, np.complex64)):
        mixed = []
        for _ in range(80):
            perm = [int(v) for v in rng.permutation(n)]
            kind = int(rng.integers(0, 8))
            if kind == 0:
                mixed.append(q.make_swap_op([perm[0]], [perm[1]]))
            elif kind == 1:
                mixed.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], GATES_1Q["dense"])))
            elif kind == 2:
                mixed.append(q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel()))
            elif kind == 3:
                mixed.append(q.make_control_op([perm[0]], q.make_swap_op([perm[1]], [perm[2]])))
            elif kind == 4:
                mixed.append(q.make_control_op(perm[:1], q.make_matrix_op([perm[1]], [1, 0, 0, cmath.rect(1, 0.7)])))
            else:
                mixed.append(q.make_matrix_op([perm[0]], GATES_1Q[["H", "T", "Rz", "X", "Y", "upper", "S"][int(rng.integers(0, 7))]]))
        for name, ops in (("c2", circuits.h_layer(n) + circuits.c2_random_circuit(n, 150, seed=28)),
                          ("qft", circuits.c3_qft(n)), ("grover", circuits.c5_grover_iteration(n)), ("mixed", mixed)):
            x = circuits.random_state(n, seed=n, dtype=dtype)
            with q.HipState(n, dtype) as st:
                st.set_option("tile", 1)
                st.upload(x)
                st.apply_ops(ops)
                want = st.download()
            before = jit_count()[0]
            with q.HipState(n, dtype) as st:
                st.set_option("tile", 1)
                st.set_option("tile_jit", 1)
                st.upload(x)
                st.apply_ops(ops)
                got = st.download()
                mid = jit_count()[0]
                st.upload(x)
                st.apply_ops(ops)  # every segment is in the cache now
                again = st.download()
                assert jit_count()[0] == mid
            assert mid > before, (name, n)
            assert np.array_equal(got, want) and np.array_equal(again, want), (name, n, dtype)
        with q.HipState(n, dtype) as st:  # a program: kernels are compiled before the capture, the graph replays them
            st.set_option("tile", 1)
            st.set_option("tile_jit", 1)
            st.upload(x)
            prog = st.compile_program(mixed)
            prog.run()
            assert prog.is_graph
            with q.HipState(n, dtype) as ref:
                ref.upload(x)
                ref.apply_ops(mixed)
                assert np.array_equal(st.download(), ref.download())
            prog.close()


def test_lds_tile_sweeps_complex64_and_programs(O):
    """tile sweeps in f32 (16-KiB tiles) are bit-identical to the f32 gate-by-gate path, and a hipGraph
    program recorded with tile = 1 replays the same result."""
    n = 14
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 150, seed=9) + circuits.c3_qft(n)[:60]
    x = circuits.random_state(n, seed=2, dtype=np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.upload(x)
        st.apply_ops(ops)
        eager = st.download()
    with q.HipState(n, np.complex64) as st:
        st.set_option("tile", 1)
        st.upload(x)
        st.apply_ops(ops)
        assert np.array_equal(st.download(), eager)
    assert np.max(np.abs(eager - O.apply_ops_in_place(n, ops, x.copy()))) < 1e-4
    x64 = circuits.random_state(n, seed=2)
    with q.HipState(n) as st:
        st.upload(x64)
        st.apply_ops(ops)
        eager64 = st.download()
    with q.HipState(n) as st:
        st.set_option("tile", 1)
        st.upload(x64)
        prog = st.compile_program(ops)
        prog.run()
        assert prog.is_graph
        assert np.array_equal(st.download(), eager64)
        prog.close()


def test_hipgraph_program_replay(O):
    """A circuit captured into a hipGraph replays bit-identically to eager application (same kernels),
    repeatedly; circuits with an out-of-place op fall back to eager transparently."""
    rng = np.random.default_rng(2)
    for n in (7, 12, 15):
        perm = [int(v) for v in rng.permutation(n)]
        circ = (circuits.h_layer(n) + circuits.c2_random_circuit(n, 100, seed=n) + circuits.c3_qft(n)[:50]
                + circuits.c5_grover_iteration(n)
                + [q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel()),
                   q.make_matrix_op(perm[:3], rand_unitary(3, rng).ravel()),
                   q.make_matrix_op(perm[2:5], np.diag(np.exp(1j * rng.uniform(0, 6, 8))).ravel()),
                   q.make_swap_op(perm[:2], perm[2:4])])
        if n >= 12:
            circ.append(q.make_matrix_op(perm[:5], rand_unitary(5, rng).ravel()))
        x = circuits.random_state(n, seed=n)
        with q.HipState(n) as st:
            st.upload(x)
            st.apply_ops(circ)
            once = st.download()
            st.apply_ops(circ)
            twice = st.download()
        with q.HipState(n) as st:
            st.upload(x)
            prog = st.compile_program(circ)
            prog.run()
            assert prog.is_graph
            assert np.array_equal(st.download(), once)
            prog.run()
            assert np.array_equal(st.download(), twice)
            prog.close()
        assert np.max(np.abs(once - O.apply_ops_in_place(n, circ, x.copy()))) <= TOL64
        # a sparse op on 6 qubits with FIVE entries in a row takes the out-of-place literal kernel: the program must stay correct
        # (eager fallback); with two entries per row it is applied in place through k_sparse_tile where the state is large enough
        # (r4) and the program is a graph again
        pos = [n - 1 - qb for qb in perm[:6]]
        kh = sum(1 for pp in pos if not (pp < 5 or pp == (11 if n >= 12 else 5)))  # the op's positions outside the wave row
        tile_form = 3 <= kh <= 7 and n >= 6 + kh + 2
        for width in (5, 2):
            rows = [[((r * 5 + 1 + 7 * e) % 64, 0.5j if e == 0 else 0.25 * (e + 1)) for e in range(width - 1)] + [(r, 2.0)] for r in range(64)]
            sp = circ[:20] + [q.make_sparse_matrix_op(perm[:6], rows)] + circ[20:40]
            with q.HipState(n) as st:
                st.upload(x)
                prog = st.compile_program(sp)
                prog.run()
                prog.run()
                assert prog.is_graph == (width == 2 and tile_form), (n, width, kh)
                got = st.download()
            want = O.apply_ops_in_place(n, sp + sp, x.copy())
            assert np.max(np.abs(got - want)) <= TOL64 * max(1.0, float(np.max(np.abs(want)))), (n, width)


def test_programs_compile_their_segments_automatically(O):
    """r5, option tile_auto (default on): apply_ops on a state with tile = 1 and tile_jit = 0 keeps the interpreter kernel; a PROGRAM
    created on that state (n >= 22) is made to be replayed and compiles its segments once at creation — wide ones, also inside
    the hipGraph — through helper processes and the disk cache.  Same helpers, same order: bit-identical to the interpreter."""
    from rustqip_amd import _ffi

    n = 22
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 120, seed=28) + circuits.c3_qft(n)[:90]
    x = circuits.random_state(n, seed=2)
    with q.HipState(n) as ref:
        ref.set_option("tile", 1)
        ref.upload(x)
        c0 = _ffi.jit_counters()
        ref.apply_ops(ops)  # the interpreter: nothing is compiled for a circuit that runs once
        assert _ffi.jit_counters()["kernels_resident_total"] == c0["kernels_resident_total"]
        want = ref.download()
    with q.HipState(n) as st:
        st.set_option("tile", 1)
        st.upload(x)
        prog = st.compile_program(ops)
        c1 = _ffi.jit_counters()
        made = c1["kernels_resident_total"] - c0["kernels_resident_total"]
        assert made >= 2, (c0, c1)  # its segments were made resident at creation (compiled here, by helpers, or found on disk)
        assert c1["compiled"] - c0["compiled"] + c1["disk_hits"] - c0["disk_hits"] >= made
        prog.run()
        assert prog.is_graph
        assert np.array_equal(st.download(), want)
        st.upload(x)
        prog.run()
        prog.run()  # replays compile nothing
        assert _ffi.jit_counters()["kernels_resident_total"] == c1["kernels_resident_total"]
        assert np.array_equal(st.download(), ref_twice(n, ops, x))
        # the state's own apply_ops still interprets (the program's options are its own)
        st.upload(x)
        st.apply_ops(ops)
        assert _ffi.jit_counters()["kernels_resident_total"] == c1["kernels_resident_total"] and np.array_equal(st.download(), want)
        prog.close()
        st.set_option("tile_auto", 0)  # switched off: a program uses the state's options as they are
        prog = st.compile_program(ops)
        assert _ffi.jit_counters()["kernels_resident_total"] == c1["kernels_resident_total"]
        st.upload(x)
        prog.run()
        assert prog.is_graph and np.array_equal(st.download(), want)
        prog.close()
    assert np.max(np.abs(want - O.apply_ops_in_place(n, ops, x.copy()))) == 0.0


def ref_twice(n, ops, x):
    with q.HipState(n) as st:
        st.upload(x)
        st.apply_ops(ops)
        st.apply_ops(ops)
        return st.download()


def test_gate_by_gate_pairs_a_line_floor_gate_with_its_neighbour(O):
    """r5, option pair_floor (default on, n >= 22): in the gate-by-gate path a gate whose selectors sit inside a wave row (T / S /
    controlled phase on a low bit, CNOT with a low control: a sweep of the whole vector for half the bytes) and the next gate go
    as ONE two-item tile sweep when they fit a tile.  Same unfused arithmetic per amplitude: IEEE-equal to one launch per gate,
    and fewer launches."""
    n = 22
    x = circuits.random_state(n, seed=6)
    rng = np.random.default_rng(3)
    low = []
    for _ in range(40):  # every kind of line-floor gate next to every kind of neighbour
        qs = [int(v) for v in rng.permutation(n)]
        lo = n - 1 - int(rng.integers(0, 6))  # a qubit whose index bit lies inside a wave row
        hi = [v for v in qs if v != lo]
        kind = int(rng.integers(0, 4))
        if kind == 0:
            low.append(q.make_matrix_op([lo], circuits.T))
        elif kind == 1:
            low.append(q.make_control_op([lo], q.make_matrix_op([hi[0]], circuits.X)))
        elif kind == 2:
            low.append(q.make_control_op([hi[0]], q.make_matrix_op([lo], [1, 0, 0, cmath.rect(1, 0.4)])))
        else:
            low.append(q.make_control_op([lo, hi[0]], q.make_matrix_op([hi[1]], circuits.H)))
        nb = int(rng.integers(0, 4))
        low.append([q.make_matrix_op([hi[2]], circuits.H), q.make_matrix_op([hi[3]], circuits.rz(0.3)),
                    q.make_control_op([hi[4]], q.make_matrix_op([hi[5]], circuits.X)), q.make_swap_op([hi[6]], [hi[7]])][nb])
    for name, ops in (("c4", circuits.c4_clifford_t(n, 120, seed=32)), ("c2", circuits.c2_random_circuit(n, 120, seed=28)), ("low", low),
                      ("qft", circuits.c3_qft(n)[:100])):
        res = {}
        for pair in (0, 1):
            with q.HipState(n) as st:
                st.set_option("pair_floor", pair)
                st.set_option("profile", 1)
                st.upload(x)
                st.apply_ops(ops)
                res[pair] = (st.download(), sum(v["launches"] for v in st.profile().values()))
        assert np.array_equal(res[0][0], res[1][0]), name
        assert res[1][1] < res[0][1], (name, res[0][1], res[1][1])  # pairs were formed
        if name in ("c4", "low"):
            assert np.array_equal(res[1][0], O.apply_ops_in_place(n, ops, x.copy())), name


def test_program_with_a_sparse_op_on_six_qubits_is_a_graph(O):
    """r4: a SparseMatrix on k >= 6 qubits with narrow rows is applied IN PLACE (k_sparse_tile), so a program that holds one
    is recorded as a hipGraph like any other (the out-of-place kernels it used to take made the program fall back to eager)."""
    n = 15
    rng = np.random.default_rng(4)
    rows = [[((r * 5 + 1) % 64, 0.5j), (r, 2.0), ((r * 11 + 3) % 64, -0.25)] for r in range(64)]
    sp = q.make_control_op([7], q.make_sparse_matrix_op([0, 1, 5, 6, 13, 14], rows))  # positions 14, 13, 9, 8 above the rows; 1, 0 inside
    circ = circuits.h_layer(n) + circuits.c2_random_circuit(n, 30, seed=1) + [sp] + circuits.c2_random_circuit(n, 30, seed=2) + [sp]
    x = circuits.random_state(n, seed=3)
    with q.HipState(n) as st:
        st.upload(x)
        prog = st.compile_program(circ)
        prog.run()
        prog.run()
        assert prog.is_graph
        got = st.download()
        prog.close()
    want = O.apply_ops_in_place(n, circ + circ, x.copy())
    assert np.max(np.abs(got - want)) <= TOL64 * max(1.0, float(np.max(np.abs(want))))
    with q.HipState(n) as st:  # and eagerly the very same bits
        st.upload(x)
        st.apply_ops(circ + circ)
        assert np.array_equal(st.download(), got)


def test_program_outliving_its_state_is_inert():
    st = q.HipState(6)
    st.init_basis(0)
    prog = st.compile_program(circuits.h_layer(6))
    prog.run()
    st.close()
    with pytest.raises(q.CircuitError, match="destroyed"):
        prog.run()
    prog.close()  # must not touch the dead state
    # and the next state works normally
    with q.HipState(6) as st2:
        st2.init_basis(0)
        st2.apply_ops(circuits.h_layer(6))
        assert abs(st2.norm_sqr() - 1) < 1e-12


def test_qft_matches_dft(O):
    """Size-independent property: the QFT circuit is the DFT matrix (bit-reversal included)."""
    n = 8
    N = 1 << n
    x = circuits.random_state(n, seed=1)
    with q.HipState(n) as st:
        st.upload(x)
        st.apply_ops(circuits.c3_qft(n))
        got = st.download()
    want = np.fft.ifft(x) * math.sqrt(N)  # QFT|j> = N^-1/2 sum_k e^{+2 pi i jk/N}|k>
    assert np.max(np.abs(got - want)) < 1e-12


# ---- builder: README CSWAP (configs[0]) --------------------------------------------------------------------
def test_cswap_readme_example(O):
    b = q.HipBuilder()
    qb = b.qubit()
    ra = b.register(3)
    rb = b.register(3)
    qb = b.h(qb)
    cb = b.condition_with(qb)
    ra, rb = cb.swap(ra, rb)
    qb = cb.dissolve()
    qb = b.h(qb)
    pre = [(e.indices, e.kind, e.param) for e in b.pipeline]
    qb, handle = b.measure(qb)
    for forced in (0, 1):
        state, measured = b.calculate_state_with_init([(ra, 0b000), (rb, 0b001)], forced_measurements=[forced])
        m, p = measured.get_measurement(handle)
        assert m == forced and abs(p - 0.5) < 1e-12
        want, res = O.run_pipeline(7, pre + [([0], "Measurement", None)], 4, forced_measurements=[forced])
        assert np.max(np.abs(state - want)) < 1e-12
        assert abs(res[0][2] - p) < 1e-12
    # pre-measurement known answer B7
    b2 = q.HipBuilder()
    b2._n, b2.pipeline = 7, b.pipeline[:-1]
    state, _ = b2.calculate_state_with_init([(ra, 0b000), (rb, 0b001)])
    expect = np.zeros(128, dtype=np.complex128)
    expect[[4, 32, 68]] = 0.5
    expect[96] = -0.5
    assert np.max(np.abs(state - expect)) < 1e-12
    # sampled (unforced) measurement returns one of the two outcomes with p = 1/2
    _, measured = b.calculate_state_with_init([(ra, 0b000), (rb, 0b001)], rng=np.random.default_rng(0))
    m, p = measured.get_measurement(handle)
    assert m in (0, 1) and abs(p - 0.5) < 1e-12


def test_builder_run_loop_uses_tile_sweeps_bit_identically(O):
    """HipBuilder (default tile = 1) on a 12-qubit circuit: same amplitudes, bit for bit, as the oracle's
    restatement of the reference run loop and as the one-sweep-per-gate builder."""
    from rustqip_amd.builder import Register

    def build(tile):
        b = q.HipBuilder(tile=tile)
        ra, rb = b.register(6), b.register(6)
        b.h(ra)
        b.cnot(Register((0,)), rb)
        b.t(rb)
        b.rz(ra, 0.37)
        b.swap_op(Register((1, 2)), Register((10, 11)))
        b.y(Register((4,)))
        b.s_dagger(Register((7,)))
        b.toffoli(Register((3, 8)), Register((5,)))
        b.h(rb)
        b.measure_stochastic(Register((2, 9)))
        _, h = b.measure(Register((6,)))
        return b, (ra, rb), h

    b1, (ra, rb), h = build(1)
    b0, _, _ = build(0)
    init = [(ra, 0b010101), (rb, 0b100001)]
    s1, m1 = b1.calculate_state_with_init(init, forced_measurements=[1])
    s0, m0 = b0.calculate_state_with_init(init, forced_measurements=[1])
    assert np.array_equal(s1, s0)
    pipe = [(e.indices, e.kind, e.param) for e in b1.pipeline]
    want, res = O.run_pipeline(12, pipe, b1.initial_index(init), forced_measurements=[1])
    assert np.max(np.abs(s1 - want)) < 1e-12
    assert np.max(np.abs(m1.get_stochastic_measurement(0) - res[0][1])) < 1e-12
    assert m1.get_measurement(h)[0] == 1 and abs(m1.get_measurement(h)[1] - res[1][2]) < 1e-12


# ---- measurement -----------------------------------------------------------------------------------------------
def test_measurement_golden_vectors():
    with q.HipState(2) as st:
        st.upload(np.array([0, 0, 1, 0], dtype=np.complex128))
        assert st.measure_prob(0, [0]) == 0.0 and st.measure_prob(1, [0]) == 1.0
        assert st.measure_prob(1, [0, 1]) == 1.0 and st.measure_prob(2, [1, 0]) == 1.0
        for r in (1e-9, 0.4, 0.99):
            assert st.soft_measure([0], r) == 1 and st.soft_measure([1], r) == 0
            assert st.soft_measure([0, 1], r) == 0b01 and st.soft_measure([1, 0], r) == 0b10
    for m, expect in ((0, [S2, S2, 0, 0]), (1, [0, 0, S2, S2])):
        with q.HipState(2) as st:
            st.upload(np.full(4, 0.5, dtype=np.complex128))
            assert list(st.measure_probs([1])) == [0.5, 0.5]
            got_m, p = st.measure([0], measured=m)
            assert got_m == m and abs(p - 0.5) < 1e-15
            assert np.max(np.abs(st.download() - np.array(expect))) < 1e-10


def test_measurement_vs_oracle(O):
    n = 13
    x = rand_state(n, 21)
    with q.HipState(n) as st:
        st.upload(x)
        assert abs(st.norm_sqr() - O.prob_magnitude(x)) < 1e-12
        for idx in ([0], [12], [3, 7], [9, 1, 4], list(range(12)), list(range(13))[::-1]):
            got = st.measure_probs(idx)
            want = O.measure_probs(n, idx, x)
            assert np.max(np.abs(got - want)) < 1e-13
            assert abs(got.sum() - 1) < 1e-12
        assert abs(st.measure_prob(5, [2, 8, 11]) - O.measure_prob(n, 5, [2, 8, 11], x)) < 1e-13
        for r in (0.001, 0.25, 0.5, 0.77, 0.9999):
            for idx in ([0, 1, 2], [12, 5]):
                assert st.soft_measure(idx, r) == O.soft_measure(n, idx, x, r)
    for forced in (0, 3, 6):
        with q.HipState(n) as st:
            st.upload(x)
            m, p = st.measure([1, 5, 10], measured=forced)
            out = np.zeros_like(x)
            wm, wp = O.measure(n, [1, 5, 10], x, out, forced=forced)
            assert (m, abs(p - wp) < 1e-13) == (wm, True)
            got = st.download()
            assert np.max(np.abs(got - out)) < 1e-12
            assert abs(st.norm_sqr() - 1) < 1e-12
    # probability-zero outcome: state untouched (measurement_ops.rs:230)
    with q.HipState(3) as st:
        st.init_basis(0)
        m, p = st.measure([0], measured=1)
        assert (m, p) == (1, 0.0)
        assert st.download()[0] == 1


# ---- full size (BASELINE sizes): size-independent properties ---------------------------------------------------
@pytest.mark.slow
@pytest.mark.parametrize("n", [28])
def test_full_size_properties(n):
    """n = 28 (4 GiB): permutations round-trip bit-exactly, a random circuit followed by its
    inverse returns to the start, the norm is preserved."""
    inv = {"H": "H", "X": "X"}
    with q.HipState(n) as st:
        st.init_basis(5)
        st.apply_ops(circuits.h_layer(n))
        assert abs(st.norm_sqr() - 1) < 1e-10
        probe0 = st.download(12345, 4096)
        assert np.allclose(np.abs(probe0), 2 ** (-n / 2), atol=1e-15)
        # X, CNOT, SWAP twice = identity, bit-exact on a window
        rng = np.random.default_rng(0)
        st.apply_ops([q.make_matrix_op([t], circuits.rz(0.1 * (t + 1))) for t in range(n)])  # make amplitudes distinct
        before = st.download(1 << 20, 1 << 16)
        perms = []
        for t in (0, 1, n // 2, n - 7, n - 2, n - 1):
            perms.append(q.make_matrix_op([t], circuits.X))
        perms.append(q.make_control_op([0], q.make_matrix_op([n - 1], circuits.X)))
        perms.append(q.make_control_op([n - 1], q.make_matrix_op([3], circuits.X)))
        perms.append(q.make_swap_op([0], [n - 1]))
        perms.append(q.make_swap_op([2, 3], [n - 3, n - 2]))
        st.apply_ops(perms + perms[::-1])
        assert np.array_equal(st.download(1 << 20, 1 << 16), before)
        # random circuit and its inverse
        ops = circuits.c2_random_circuit(n, 48, seed=n)
        inverse = []
        for op in reversed(ops):
            if op.kind == "Control":
                inverse.append(op)
            else:
                d = op.data
                inverse.append(q.make_matrix_op(op.indices, np.conj(d.reshape(2, 2).T).ravel()))
        st.apply_ops(ops)
        assert abs(st.norm_sqr() - 1) < 1e-10
        st.apply_ops(inverse)
        after = st.download(1 << 20, 1 << 16)
        assert np.max(np.abs(after - before)) < 1e-12
        assert abs(st.norm_sqr() - 1) < 1e-10


@pytest.mark.slow
def test_full_size_qft_and_grover_properties():
    """BASELINE configs[2] and [4] at n = 28 on one GPU, checked by size-independent properties:
    QFT of a basis state has the closed form N^-1/2 exp(2 pi i j k / N); QFT followed by its inverse is
    the identity; one Grover iteration takes the marked amplitude to sin(3 theta)."""
    n = 28
    N = 1 << n
    j = 0b1011001110001111000011111010  # 28-bit basis state
    qft = circuits.c3_qft(n)
    with q.HipState(n) as st:
        st.init_basis(j)
        st.apply_ops(qft)
        assert abs(st.norm_sqr() - 1) < 1e-10
        for k0 in (0, 1, 12345, N // 2 + 77, N - 4096):
            got = st.download(k0, 4096)
            k = np.arange(k0, k0 + 4096, dtype=np.int64)
            phase = ((j * k) % N).astype(np.float64)  # j*k < 2^56 is exact in int64
            want = np.exp(2j * np.pi * phase / N) / math.sqrt(N)
            assert np.max(np.abs(got - want)) < 1e-13, k0  # amplitudes are 2^-14
        inverse = []
        for op in reversed(qft):
            if op.kind == "Swap":
                inverse.append(op)
            elif op.kind == "Control":
                d = op.inner.data
                inverse.append(q.make_control_op(op.indices[:1], q.make_matrix_op(op.indices[1:], np.conj(d))))
            else:
                inverse.append(op)  # H
        st.apply_ops(inverse)
        back = st.download(j - 5, 16)
        expect = np.zeros(16, dtype=np.complex128)
        expect[5] = 1
        assert np.max(np.abs(back - expect)) < 1e-10
        assert abs(st.norm_sqr() - 1) < 1e-10
    theta = math.asin(2 ** (-n / 2))
    for dense_k3 in (False, True):
        with q.HipState(n) as st:
            st.init_basis(0)
            st.apply_ops(circuits.h_layer(n) + circuits.c5_grover_iteration(n, dense_k3=dense_k3))
            a0 = st.download(0, 2)
            assert abs(abs(a0[0]) - math.sin(3 * theta)) < 1e-12
            assert abs(abs(a0[1]) - math.cos(3 * theta) / math.sqrt(N - 1)) < 1e-12
            assert abs(st.norm_sqr() - 1) < 1e-10


def _permuted(n, pi, x):
    j = np.arange(1 << n, dtype=np.uint64)
    src = np.zeros_like(j)
    for dbit in range(n):
        src |= ((j >> np.uint64(dbit)) & np.uint64(1)) << np.uint64(pi[dbit])
    return x[src.astype(np.int64)]


def test_bit_permutation_in_one_sweep():
    """k_permute_bits: new[j] = old[src(j)] for any permutation of the index bits (what a run of Swap ops composes to),
    IEEE-equal, both precisions (the packed 16-byte view of Complex<f32> when bit 0 stays, 8-byte elements otherwise),
    states smaller than a tile included."""
    rng = np.random.default_rng(21)
    for dtype in (np.complex128, np.complex64):
        for n in (3, 9, 10, 11, 12, 13, 16, 20):
            x = rand_state(n, n, dtype)
            perms = [list(range(n))[::-1], list(range(1, n)) + [0], [n - 1] + list(range(1, n - 1)) + [0]]
            perms += [[int(v) for v in rng.permutation(n)] for _ in range(4)]
            perms += [[0] + [1 + int(v) for v in rng.permutation(n - 1)] for _ in range(2)]  # bit 0 fixed: packed f32 view
            perms += [list(range(n))]  # identity: no launch
            with q.HipState(n, dtype) as st:
                for pi in perms:
                    st.upload(x)
                    st.permute_bits(pi)
                    assert np.array_equal(st.download(), _permuted(n, pi, x)), (dtype, n, pi)
                # a chain of permutations on the resident state (the buffers alternate)
                st.upload(x)
                want = x
                for pi in perms[:5]:
                    st.permute_bits(pi)
                    want = _permuted(n, pi, want)
                assert np.array_equal(st.download(), want)
                with pytest.raises(q.CircuitError):
                    st.permute_bits([0] * n if n > 1 else [1])


def test_tile_schedule_sends_runs_of_swaps_through_the_permutation_sweep(O):
    """tile >= 1: QFT's closing bit reversal (and any run of uncontrolled Swap ops no single segment can hold) is ONE
    out-of-place sweep.  Swaps only move amplitudes, so the state stays IEEE-equal to the gate-by-gate path."""
    n = 20
    x = rand_state(n, 5)
    ops = circuits.c3_qft(n)
    want = O.apply_ops_in_place(n, ops, x.copy())
    with q.HipState(n) as st:
        st.upload(x)
        st.apply_ops(ops)
        plain = st.download()
        assert np.max(np.abs(plain - want)) <= TOL64
        for opts in ({"tile": 1}, {"tile": 1, "tile_jit": 1}):
            for k, v in opts.items():
                st.set_option(k, v)
            st.upload(x)
            st.set_option("profile", 1)
            st.profile_reset()
            st.apply_ops(ops)
            prof = st.profile()
            st.set_option("profile", 0)
            assert prof["k_permute_bits"]["launches"] == 1, prof
            assert np.array_equal(st.download(), plain), opts
        st.set_option("tile_jit", 0)
        # swaps only: bit reversal + a second run, f32 as well
        rev = [q.make_swap_op([i], [n - 1 - i]) for i in range(n // 2)] + [q.make_swap_op([0, 3, 5], [19, 7, 11]), q.make_swap_op([2], [9])]
        st.upload(x)
        st.apply_ops(rev)
        assert np.array_equal(st.download(), O.apply_ops_in_place(n, rev, x.copy()))
        # a program whose schedule holds a permutation sweep stays eager and stays right
        from rustqip_amd.state import HipProgram

        prog = HipProgram(st, ops)
        for _ in range(2):
            st.upload(x)
            prog.run()
            assert not prog.is_graph and np.array_equal(st.download(), plain)
        prog.close()
    xf = rand_state(n, 6, np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.set_option("tile", 1)
        st.upload(xf)
        st.apply_ops(rev)
        assert np.array_equal(st.download(), O.apply_ops_in_place(n, rev, xf.copy()))


@pytest.mark.slow
def test_full_size_qft_through_tile_sweeps_and_the_permutation_sweep():
    """configs[2] at n = 28 with tile = 1 (run-time-compiled segments): closed form of QFT|j>, then QFT^-1 back to |j>."""
    n = 28
    N = 1 << n
    j = 0b0110100111000111100001111101
    qft = circuits.c3_qft(n)
    with q.HipState(n) as st:
        st.set_option("tile", 1)
        st.set_option("tile_jit", 1)
        st.init_basis(j)
        st.apply_ops(qft)
        for k0 in (0, 3, 54321, N // 2 + 99, N - 4096):
            got = st.download(k0, 4096)
            k = np.arange(k0, k0 + 4096, dtype=np.int64)
            want = np.exp(2j * np.pi * ((j * k) % N).astype(np.float64) / N) / math.sqrt(N)
            assert np.max(np.abs(got - want)) < 1e-13, k0
        inverse = []
        for op in reversed(qft):
            if op.kind == "Control":
                inverse.append(q.make_control_op(op.indices[:1], q.make_matrix_op(op.indices[1:], np.conj(op.inner.data))))
            else:
                inverse.append(op)
        st.apply_ops(inverse)  # starts with the run of swaps: one permutation sweep
        back = st.download(j - 5, 16)
        expect = np.zeros(16, dtype=np.complex128)
        expect[5] = 1
        assert np.max(np.abs(back - expect)) < 1e-10
        assert abs(st.norm_sqr() - 1) < 1e-10


def test_tile_relabel_is_bit_identical(O):
    """option tile_relabel: the tile scheduler keeps a logical -> physical map of the qubits (soonest-needed qubits on index
    bits 0..5 through in-tile swaps, Swap ops as label exchanges, one closing bit-permutation sweep).  Only moves are
    added and no gate changes its place in the plain schedule's order, so the state is IEEE-equal to tile = 1 / 2 without it
    (and to the gate-by-gate path for tile = 1); interpreter and run-time-compiled segments, both precisions."""
    rng = np.random.default_rng(77)
    for n, gates in ((13, 150), (16, 220), (20, 300)):
        ops = circuits.c2_random_circuit(n, gates, seed=n)
        ops = ops[: gates // 2] + [q.make_swap_op([1], [n - 2]), q.make_swap_op([0, 2], [n - 1, 5])] + ops[gates // 2:]
        x = rand_state(n, n)
        want = O.apply_ops_in_place(n, ops, x.copy())
        with q.HipState(n) as st:
            st.upload(x)
            st.apply_ops(ops)
            plain = st.download()
            assert np.max(np.abs(plain - want)) <= TOL64
            for tile in (1, 2):
                st.set_option("tile", tile)
                st.upload(x)
                st.apply_ops(ops)
                base = st.download()
                if tile == 1:
                    assert np.array_equal(base, plain)
                for relabel in (1, 2):
                    for jit in (0, 1):
                        if jit and n != 16:
                            continue
                        st.set_option("tile_relabel", relabel)
                        st.set_option("tile_jit", jit)
                        st.upload(x)
                        st.apply_ops(ops)
                        got = st.download()
                        if tile == 1:
                            assert np.array_equal(got, plain), (n, tile, relabel, jit)
                        else:  # tile = 2 hoists gates: the relabelled plan may group them differently (1e-12 bar, as tile = 2 itself)
                            assert np.max(np.abs(got - want)) <= TOL64, (n, tile, relabel, jit)
                st.set_option("tile_relabel", 0)
                st.set_option("tile_jit", 0)
            st.set_option("tile", 0)
    n = 16
    ops = circuits.c4_clifford_t(n, 200, seed=9) + circuits.c3_qft(n)[:60]
    xf = rand_state(n, 3, np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.set_option("tile", 1)
        st.upload(xf)
        st.apply_ops(ops)
        base = st.download()
        st.set_option("tile_relabel", 2)
        st.upload(xf)
        st.apply_ops(ops)
        assert np.array_equal(st.download(), base)
    del rng


@pytest.mark.slow
def test_full_size_tile_relabel_saves_sweeps_and_changes_nothing():
    """configs[1] at n = 28: the relabelled plan needs fewer sweeps (profile) and leaves the very same state, compared
    on windows of two resident states (bottom, top and places in between)."""
    n = 28
    N = 1 << n
    ops = circuits.c2_random_circuit(n, 256, seed=28)
    init = circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)]
    with q.HipState(n) as a, q.HipState(n) as b:
        launches = []
        for st, relabel in ((a, 0), (b, 1)):
            st.init_basis(5)
            st.apply_ops(init)
            st.set_option("tile", 1)
            st.set_option("tile_relabel", relabel)
            st.set_option("profile", 1)
            st.profile_reset()
            st.apply_ops(ops)
            prof = st.profile()
            st.set_option("profile", 0)
            launches.append(sum(v["launches"] for v in prof.values()))
            if relabel:
                assert prof.get("k_permute_bits", {}).get("launches", 0) == 1, prof
        assert launches[1] < launches[0], launches
        for off in (0, 1 << 16, 123456789 & ~0xFFFF, N // 2 - (1 << 15), N - (1 << 16)):
            assert np.array_equal(a.download(off, 1 << 16), b.download(off, 1 << 16)), off
        assert abs(b.norm_sqr() - 1) < 1e-10


def test_window_compare_at_n24(O):
    """Full-vector compare against the oracle at n = 24 on a prefix of configs[1]."""
    n = 24
    ops = circuits.h_layer(n)[:6] + circuits.c2_random_circuit(n, 10, seed=28)
    x = rand_state(n, 24)
    with q.HipState(n) as st:
        st.upload(x)
        st.apply_ops(ops)
        got = st.download()
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= TOL64


# ---- full size, directly against the oracle: closed sub-cubes of the index space (oracle/window_parity.py) --------
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


@pytest.mark.slow
@pytest.mark.parametrize("n", [30, 32])  # (r5: n = 28 dropped — 30 is the timed size, 32 the first 2-D grids; the suite's time limit)
def test_full_size_oracle_windows(O, n):
    """The benchmarked sizes (n = 30 is bench.py's workload; n = 32 is where streaming launches first need a second
    grid dimension) compared with the ORACLE, gate by gate, on a seeded product state whose amplitudes are pairwise
    distinct: >= 4 closed sub-cubes of >= 2^16 rows per gate, always including the bottom and the top of the index
    space.  f64 gate-by-gate kernels are bit-equal (array_equal); matrix-core gates are held to 1e-12."""
    from oracle import window_parity as W

    rng = np.random.default_rng(n)
    ops0, vecs = W.product_state_ops(n, seed=n)
    c2 = circuits.c2_random_circuit(n, 256, seed=28)
    N = 1 << n
    with q.HipState(n) as st:
        st.init_basis(0)
        st.apply_ops(ops0)
        for off in (0, 12345, N // 2 - 777, N - (1 << 16)):
            got = st.download(off, 1 << 16)
            want = W.product_state_window(n, vecs, off, 1 << 16)
            assert np.allclose(got, want, rtol=1e-12, atol=0), off
        assert abs(st.norm_sqr() - 1) < 1e-10
        # the first gates of the benchmarked circuit, one launch per gate
        n_gbg = 40 if n <= 30 else 32
        agg = W.check_circuit(st, n, c2[:n_gbg], O, gate_by_gate=True, seed=1)
        assert agg["gates"] == n_gbg and agg["skipped"] == 0 and agg["rows"] >= n_gbg * 4 * (1 << 16)
        assert agg["bit_equal"] and agg["max_abs_delta"] == 0.0, agg
        for name, op, exact in _special_gates(n, rng):
            r = W.check_ops(st, n, [op], O, bases=W.default_bases(n, seed=zlib.crc32(name.encode()) % 1000))
            assert r is not None, name
            if exact:
                assert r["bit_equal"], (name, r)
            else:
                assert r["max_abs_delta"] <= TOL64, (name, r)
        # LDS-resident multi-gate sweeps (tile = 1: IEEE-equal) on the following gates of the same circuit, in
        # chunks whose sub-cubes stay small; the chunk is what the tile scheduler sees
        st.set_option("tile", 1)
        st.set_option("profile", 1)
        st.profile_reset()
        n_tile = 96 if n <= 30 else 48
        agg = W.check_circuit(st, n, c2[40:40 + n_tile], O, gate_by_gate=False, seed=2, bases_per_step=2)
        prof = st.profile()
        assert agg["gates"] == n_tile and agg["skipped"] == 0
        assert prof.get("k_tile_passes", {}).get("launches", 0) >= 3, prof  # multi-gate sweeps really ran
        assert agg["max_abs_delta"] == 0.0, agg  # only a -0 may differ from the gate-by-gate path
        if n == 30:
            # the same sweeps as kernels compiled at run time for each segment (option tile_jit): still IEEE-equal
            st.set_option("tile_jit", 1)
            agg = W.check_circuit(st, n, c2[136:200], O, gate_by_gate=False, seed=4, bases_per_step=2)
            st.set_option("tile_jit", 0)
            assert agg["gates"] == 64 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
            # the scheduler relabelling the qubits (option tile_relabel = 2: unconditionally, so every chunk goes through
            # in-tile swaps, label exchanges and the closing bit-permutation sweep): still IEEE-equal to the oracle
            st.set_option("tile_relabel", 2)
            qswap = [q.make_swap_op([3], [n - 2]), q.make_swap_op([n - 9], [0])]
            agg = W.check_circuit(st, n, c2[200:232] + qswap + c2[232:256], O, gate_by_gate=False, seed=5, bases_per_step=2)
            st.set_option("tile_relabel", 0)
            assert agg["gates"] == 58 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
            assert st.profile().get("k_permute_bits", {}).get("launches", 0) >= 1
            # configs[4], dense k = 3 variant (the 8x8 gates ride in the sweeps as passes of their own three bits), with
            # dense k = 5 / k = 4 gates in between: matrix-core launches mixed with tile sweeps
            g = circuits.c5_grover_iteration(n, dense_k3=True)
            k5 = q.make_matrix_op([n - 1, n - 2, 3, n - 4, n - 5], rand_unitary(5, rng).ravel())
            k4 = q.make_matrix_op([n - 1, n - 3, 2, n - 6], rand_unitary(4, rng).ravel())
            g = g[:40] + [k5] + g[40:100] + [k4] + g[100:]
            st.profile_reset()
            agg = W.check_circuit(st, n, g, O, gate_by_gate=False, seed=3, bases_per_step=2)
            assert agg["gates"] >= len(g) - 2 and agg["max_abs_delta"] <= TOL64, agg
            assert st.profile().get("k_gate_kq_mfma", {}).get("launches", 0) >= 2, st.profile()
        st.set_option("tile", 0)
        st.set_option("profile", 0)
        assert abs(st.norm_sqr() - 1) < 1e-9


@pytest.mark.slow
def test_full_size_oracle_windows_n33(O):
    """n = 33 on ONE GPU (128 GiB: the size BASELINE configs[4] shards over 8; VERDICT r3 weak item 3: until now only its
    norm was checked): gate by gate and through tile sweeps against the oracle on closed sub-cubes, bottom and top of the
    2^33 index space included.  There is no room for a twin state; the whole-vector guard is the closed-form marginals of
    the seeded product state (single-qubit gates keep it a product state).  Nothing here may take the out-of-place path."""
    from oracle import window_parity as W

    n = 33
    ops0, vecs = W.product_state_ops(n, seed=n)
    head = circuits.c2_random_circuit(n, 64, seed=28, single_only=True)
    c2 = circuits.c2_random_circuit(n, 128, seed=28)
    N = 1 << n
    with q.HipState(n) as st:
        st.init_basis(0)
        st.apply_ops(ops0)
        for off in (0, N // 3, N - (1 << 16)):
            assert np.allclose(st.download(off, 1 << 16), W.product_state_window(n, vecs, off, 1 << 16), rtol=1e-12, atol=0), off
        guard = W.ProductGuard(n, vecs)
        assert guard.check(st) <= 1e-11
        agg = W.check_circuit(st, n, head[:16], O, gate_by_gate=True, seed=1)
        assert agg["gates"] == 16 and agg["skipped"] == 0 and agg["bit_equal"] and agg["rows"] >= 16 * 4 * (1 << 16), agg
        for op in head[:16]:
            guard.apply(op)
        assert guard.check(st) <= 1e-11, guard.worst_rel
        st.set_option("tile", 1)
        agg = W.check_circuit(st, n, head[16:64], O, gate_by_gate=False, seed=2, bases_per_step=2)
        assert agg["gates"] == 48 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
        for op in head[16:64]:
            guard.apply(op)
        assert guard.check(st) <= 1e-11, guard.worst_rel
        # the configs[1] mix (CNOTs: half sweeps whose controls sit on every kind of position), gate by gate and as sweeps
        agg = W.check_circuit(st, n, c2[:16], O, gate_by_gate=True, seed=3, bases_per_step=2)
        st.set_option("tile", 0)
        agg2 = W.check_circuit(st, n, c2[16:32], O, gate_by_gate=True, seed=4, bases_per_step=2)
        assert agg["skipped"] == 0 and agg2["skipped"] == 0 and agg["max_abs_delta"] == 0.0 and agg2["bit_equal"], (agg, agg2)
        assert abs(st.norm_sqr() - 1) < 1e-9


@pytest.mark.slow
def test_full_size_oracle_windows_complex64(O):
    """SURVEY.md §8 row f3 at the benchmarked size: a Complex<f32> state at n = 30 (8 GiB; the packed 16-byte view and the
    8-byte kernels both occur) against the f32 ORACLE on closed sub-cubes, gate by gate and through tile sweeps with the
    qubits relabelled.  Both sides compute in unfused f32, so the comparison is held to bit equality."""
    from oracle import window_parity as W

    n = 30
    ops0, vecs = W.product_state_ops(n, seed=n)
    c2 = circuits.c2_random_circuit(n, 256, seed=28)
    with q.HipState(n, np.complex64) as st:
        st.init_basis(0)
        st.apply_ops(ops0)
        for off in (0, (1 << n) - (1 << 16)):
            got = st.download(off, 1 << 16)
            want = W.product_state_window(n, vecs, off, 1 << 16)
            assert np.allclose(got, want, rtol=2e-5, atol=0), off
        agg = W.check_circuit(st, n, c2[:24], O, gate_by_gate=True, seed=21)
        assert agg["gates"] == 24 and agg["skipped"] == 0 and agg["rows"] >= 24 * 4 * (1 << 16)
        assert agg["bit_equal"] and agg["max_abs_delta"] == 0.0, agg
        st.set_option("tile", 1)
        agg = W.check_circuit(st, n, c2[24:88], O, gate_by_gate=False, seed=22, bases_per_step=2)
        assert agg["gates"] == 64 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
        st.set_option("tile_relabel", 2)
        agg = W.check_circuit(st, n, c2[88:152] + [q.make_swap_op([2], [n - 3])], O, gate_by_gate=False, seed=23, bases_per_step=2)
        assert agg["gates"] == 65 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
        # r5: the f32 WIDE tiles — what bench.py times as extras.complex64_n30.mixed_tile1_jit_wide — at the timed
        # size against the f32 oracle: run-time-compiled 13-bit register-resident segments, circuit order, then relabelled
        st.set_option("tile_relabel", 0)
        st.set_option("tile_jit", 1)
        st.set_option("tile_wide", 1)
        agg = W.check_circuit(st, n, c2[152:216], O, gate_by_gate=False, seed=24, bases_per_step=2)
        assert agg["gates"] == 64 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
        st.set_option("tile_relabel", 2)
        agg = W.check_circuit(st, n, c2[216:256], O, gate_by_gate=False, seed=25, bases_per_step=2)
        assert agg["gates"] == 40 and agg["skipped"] == 0 and agg["max_abs_delta"] == 0.0, agg
        assert abs(st.norm_sqr() - 1) < 1e-4


# ---- N > 1 on one GPU: virtual shards (real kernels, host-staged exchange) and RCCL plumbing ----------------
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


def test_sharded_virtual_shards_on_one_gpu():
    """world_size 2 on ONE GPU (world 4 is covered on CPU by tests/test_distributed_cpu.py)."""
    out = _run_dist(2, [])
    assert out.count("ok n=") == 4
    assert "ok fault: a failed exchange poisons the handle" in out and "ok pieces:" in out
    assert "ok fold: the remap's gather rides in the preceding tile sweep" in out
    assert "ok overlap: the exchange in slices beside the neighbouring tile sweeps changes nothing" in out
    assert "ok pair_floor on shards" in out
    assert out.count("samples differ from the reference's scan") == 2


@pytest.mark.slow
def test_sharded_state_against_the_oracle_at_bench_shard_size():
    """2 ranks x 2^28 amplitudes on ONE GPU (n = 29): the sharded path — localized ops, tile sweeps on the shards, k_pack_bits,
    the k_permute_bits route of a pack that gathers index bit 0, 2-D grids — checked against the oracle on closed sub-cubes of
    the LOGICAL index space read through the layout, with a twin sharded state on the literal kernel compared over all 2^29
    amplitudes after every step and closed-form marginals through qip_hip_dist_measure_probs (tests/dist_worker_parity_gpu.py)."""
    import json

    out = _run_dist(2, ["--n-local", "28", "--quick"], worker="dist_worker_parity_gpu.py", timeout=1800)
    res = json.loads([l for l in out.splitlines() if l.startswith("SHARDED_PARITY ")][-1][len("SHARDED_PARITY "):])
    assert res["all_legs_ok"] and res["n"] == 29 and res["rows_checked"] >= 10**7, res
    assert res["whole_vector"]["amplitudes_not_equal_in_IEEE_legs"] == 0 and res["bit_equal"], res


def test_sharded_rccl_plumbing_world1():
    out = _run_dist(1, ["--nccl"])
    assert out.count("ok n=") == 4


@pytest.mark.slow
def test_bench_multi_rank_code_path_on_one_gpu():
    """`python bench.py --gpus 2` as a PLAIN command (it re-launches itself as two ranks under torch.distributed.run):
    sharded state, plan/run_plan, remap, max-over-ranks timing, JSON line with the BASELINE configs[3]/[4] legs — with
    two ranks sharing the one GPU through the gloo / host-staged test hook."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--n-local", "20", "--gates", "64", "--dist-overlap", "4"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, QIP_BENCH_DIST_BACKEND="gloo"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["n_qubits"] == 21 and line["scaling"] == "weak"
    assert abs(line["norm_sqr_after"] - 1) < 1e-10
    # (the look-ahead keeps the qubit whose next H is farthest on the rank bit, and X gates there only rename the ranks:
    # a short headline circuit may need no exchange at all — the legs below and the parity leg do)
    assert line["value"] > 0 and line["roofline"]["kernel"].startswith("k_") and line["comm"]["remaps"] >= 0
    ex = line["extras"]
    for name in ("configs3_clifford_t_n21", "configs4_grover_iteration_n21", "configs4_grover_dense_k3_n21",
                 "configs1_mixed_n21", "headline_tiled_mode1", "configs3_clifford_t_tiled_mode1", "configs1_mixed_tiled_mode1",
                 "configs1_mixed_tiled_mode1_jit_wide", "configs1_mixed_tiled_mode1_jit_wide_overlap", "configs1_mixed_tiled_mode1_overlap"):
        assert "error" not in ex[name] and ex[name]["ops_per_s"] > 0, (name, ex[name])
    assert sum(ex[name]["comm_over_reps"]["remaps"] for name in ("configs3_clifford_t_n21", "configs4_grover_iteration_n21", "configs1_mixed_n21")) >= 1
    assert abs(ex["norm_sqr_end"] - 1) < 1e-9
    par = line["parity"]  # the sharded path against the oracle, inside the bench run itself
    assert "error" not in par and par["world"] == 2 and par["remaps_exercised"] >= 1 and par["max_abs_delta"] <= 1e-12, par
    # r4: the sharded state is checked at the size it was timed at (sub-cubes through the layout + twin + marginals), and the
    # verdict is a top-level field the run's exit status follows
    assert line["parity_ok"] is True and par["all_legs_ok"] and par["n"] == 21 and par["small_full_vector"]["ok"], par
    assert par["whole_vector"]["amplitudes_not_equal_in_IEEE_legs"] == 0 and par["packs_via_permute_bits"] >= 1, par


@pytest.mark.slow
def test_bench_fails_when_parity_fails(tmp_path):
    """a parity failure must be fatal — rc != 0, value null, parity_ok false at top level.  The failure is
    provoked through the checker's side only (QIP_BENCH_SABOTAGE_PARITY perturbs what the ORACLE is fed), never the product."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--n-local", "20", "--gates", "64",
           "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, QIP_BENCH_SABOTAGE_PARITY="1"))
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert res.returncode != 0 and line["value"] is None and line["parity_ok"] is False, (res.returncode, line["value"], line["parity_ok"])
    assert "PARITY FAILED" in res.stderr
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert res.returncode == 0 and line["value"] > 0 and line["parity_ok"] is True


def test_circuit_replay_python_and_cpp_cli(O, tmp_path):
    """a "qipc 1" file replayed by rustqip_amd.replay and by tools/qip_replay (C++ host
    mirror) gives the oracle's amplitudes / probabilities; the two replays print identical numbers."""
    import subprocess

    from rustqip_amd import replay

    n = 9
    rng = np.random.default_rng(5)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 60, seed=3) + circuits.c3_qft(n)[:30]
    ops.append(q.make_sparse_matrix_op([1, 4], [[(0, 1)], [(2, 1j)], [(1, -1)], [(3, cmath.rect(1, 0.4))]]))
    ops.append(q.make_matrix_op([2, 7, 0], rand_unitary(3, rng).ravel()))
    circ = replay.Circuit(n, 5, ops[:50] + [replay.Probs([0, 3, 8])] + ops[50:] + [replay.Probs([1, 2])])
    path = tmp_path / "c.qipc"
    replay.dump(str(path), circ)
    x = np.zeros(1 << n, dtype=np.complex128)
    x[5] = 1
    want = O.apply_ops_in_place(n, ops, x.copy())
    mid = O.apply_ops_in_place(n, ops[:50], x.copy())
    results, st = replay.run(replay.load(str(path)), tile=1)
    try:
        got = st.download()
    finally:
        st.close()
    assert np.max(np.abs(got - want)) <= TOL64
    assert np.max(np.abs(results[0] - O.measure_probs(n, [0, 3, 8], mid))) <= TOL64
    assert np.max(np.abs(results[1] - O.measure_probs(n, [1, 2], want))) <= TOL64
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "qip_replay"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ROOT, "tools", "qip_replay"), "--tile", "1", "--amps", "16", str(path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    probs = [np.array([float(t) for t in ln.split()[1:]]) for ln in lines if ln.startswith("probs")]
    assert len(probs) == 2 and np.array_equal(probs[0], results[0]) and np.array_equal(probs[1], results[1])
    amps = np.array([complex(float(ln.split()[2]), float(ln.split()[3])) for ln in lines if ln.startswith("amp ")])
    assert np.array_equal(amps, got[:16])  # same library, same launches: identical to the last bit


@pytest.mark.parametrize("n, seed", [(8, 1), (12, 2), (14, 3), (15, 4)])
def test_tile_sweeps_fuzz_every_gate_shape(O, n, seed):
    """Seeded fuzz over the shapes a tile segment can hold — 1-qubit gates of every zero pattern with 0..4
    controls, (multi-)controlled diagonal gates, (controlled) swaps — mixed with ops that are not tileable.
    tile = 1 must equal the gate-by-gate path under IEEE ==, both must equal the oracle (gate by gate: bit for
    bit on f64), tile = 2 / programs to 1e-12.  n > 11 puts targets and controls outside the tile as well."""
    rng = np.random.default_rng(seed)
    names = list(GATES_1Q)
    ops = []
    for _ in range(160):
        perm = [int(v) for v in rng.permutation(n)]
        shape = int(rng.integers(0, 8))
        nc = int(rng.integers(0, min(5, n - 2)))
        if shape <= 3:  # dense / diagonal 1-qubit gate, any number of controls
            g = q.make_matrix_op([perm[0]], GATES_1Q[names[int(rng.integers(0, len(names)))]])
            ops.append(q.make_control_op(perm[1:1 + nc], g) if nc else g)
        elif shape == 4:  # controlled phase with a random angle
            g = q.make_matrix_op([perm[0]], [1, 0, 0, cmath.rect(1, float(rng.uniform(0, 6.28)))])
            ops.append(q.make_control_op(perm[1:2 + nc], g))
        elif shape == 5:  # (controlled) swap
            g = q.make_swap_op([perm[0]], [perm[1]])
            ops.append(q.make_control_op(perm[2:2 + nc], g) if nc else g)
        elif shape == 6:  # dense 2- / 3-qubit gate with 0..2 controls (tileable) / 2+2 swap (not)
            pick = int(rng.integers(0, 4))
            if pick <= 1:
                g = q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel())
                ops.append(q.make_control_op(perm[2:2 + min(nc, 2)], g) if nc and pick else g)
            elif pick == 2:  # dense 3-qubit gate with 0..2 controls: a tile pass of its own three bits
                g = q.make_matrix_op(perm[:3], rand_unitary(3, rng).ravel())
                ops.append(q.make_control_op(perm[3:3 + min(nc, 2)], g) if nc and rng.integers(0, 2) else g)
            else:
                ops.append(q.make_swap_op(perm[:2], perm[2:4]))
        else:  # sparse (generic gather path)
            ops.append(q.make_sparse_matrix_op([perm[0]], [[(1, 1j)], [(0, -1j)]]))
    x = circuits.random_state(n, seed=seed)
    want = O.apply_ops_in_place(n, ops, x.copy())
    with q.HipState(n) as st:
        st.set_option("mfma", 0)  # dense k = 3 on the VALU: bit-equal to the oracle (the matrix cores are an fma chain)
        st.upload(x)
        st.apply_ops(ops)
        eager = st.download()
    assert np.array_equal(eager, want)
    for mode in (1, 2):
        with q.HipState(n) as st:
            st.set_option("mfma", 0)
            st.set_option("tile", mode)
            st.upload(x)
            st.apply_ops(ops)
            got = st.download()
        if mode == 1:
            assert np.array_equal(got, eager), (n, seed)
        else:
            assert np.max(np.abs(got - eager)) <= TOL64 * max(1.0, float(np.max(np.abs(eager)))), (n, seed)
    with q.HipState(n) as st:
        st.set_option("mfma", 0)
        st.set_option("tile", 1)
        st.upload(x)
        prog = st.compile_program(ops)
        prog.run()
        assert np.array_equal(st.download(), eager)
        prog.close()
    if n in (12, 14):  # run-time-compiled segments, and the scheduler relabelling the qubits on top: still IEEE-equal
        with q.HipState(n) as st:
            st.set_option("mfma", 0)
            st.set_option("tile", 1)
            st.set_option("tile_jit", 1)
            for relabel in (0, 2):
                st.set_option("tile_relabel", relabel)
                st.upload(x)
                st.apply_ops(ops)
                assert np.array_equal(st.download(), eager), (n, seed, relabel)
    x32 = x.astype(np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.set_option("mfma", 0)  # (a dense 3-qubit gate is a tile item now: compare with the unfused VALU form)
        st.upload(x32)
        st.apply_ops(ops)
        e32 = st.download()
    with q.HipState(n, np.complex64) as st:
        st.set_option("mfma", 0)
        st.set_option("tile", 1)
        st.upload(x32)
        st.apply_ops(ops)
        assert np.array_equal(st.download(), e32), (n, seed)


def test_resource_and_option_errors_are_reported_not_fatal():
    """A state that cannot fit in HBM, an unknown option, an op on more qubits than the state has: each is a
    status + message (QipHipError / CircuitError), and the library keeps working afterwards."""
    with pytest.raises(q.QipHipError) as e:
        q.HipState(40)  # 16 TiB: hipMalloc fails cleanly
    assert "40-qubit" in str(e.value)
    with pytest.raises((q.QipHipError, q.CircuitError)):
        q.HipState(41)
    with q.HipState(6) as st:
        with pytest.raises((q.QipHipError, q.CircuitError)) as e:
            st.set_option("no_such_option", 1)
        assert "no_such_option" in str(e.value)
        with pytest.raises((q.QipHipError, q.CircuitError)):
            st.apply_op(q.make_matrix_op([6], GATES_1Q["H"]))  # qubit 6 of a 6-qubit state
        with pytest.raises((q.QipHipError, q.CircuitError)):
            st.measure_probs([0, 0])
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(6))  # still usable
        assert abs(st.norm_sqr() - 1.0) < 1e-12
        assert np.allclose(st.measure_probs([1, 4]), 0.25)


def test_two_states_side_by_side_copy_and_whole_vector_diff():
    """qip_hip_state_copy_from / qip_hip_state_max_abs_diff: the primitives of the whole-vector guard"""
    n = 14
    x = circuits.random_state(n, 3)
    with q.HipState(n) as a, q.HipState(n) as b:
        a.upload(x)
        b.copy_from(a)
        assert np.array_equal(b.download(), x)
        assert a.max_abs_diff(b) == (0.0, 0)
        y = x.copy()
        y[777] += 1e-9
        y[(1 << n) - 1] = -y[(1 << n) - 1]
        b.upload(y)
        worst, differ = a.max_abs_diff(b)
        assert differ == 2 and abs(worst - 2 * abs(x[-1])) < 1e-15
        y = x.copy()
        y[5] = complex(float("nan"), 0.0)
        b.upload(y)
        worst, differ = a.max_abs_diff(b)
        assert differ == 1 and worst != worst
        with q.HipState(n, np.complex64) as c:
            with pytest.raises(q.CircuitError):
                a.max_abs_diff(c)
            with pytest.raises(q.CircuitError):
                c.copy_from(a)
    with q.HipState(n, np.complex64) as a, q.HipState(n, np.complex64) as b:
        a.upload(x.astype(np.complex64))
        b.copy_from(a)
        a.apply_op(q.make_matrix_op([n - 1], circuits.X))
        worst, differ = a.max_abs_diff(b)
        assert differ == 1 << n and worst > 0
        b.apply_op(q.make_matrix_op([n - 1], circuits.X))
        assert a.max_abs_diff(b) == (0.0, 0)


@pytest.mark.slow
@pytest.mark.parametrize("n", [26])
def test_full_size_every_timed_leg_with_whole_vector_guard(O, n):
    """What bench.py's parity block does at n = 30, as a test at n = 26 (r5: was 28 — the n = 30 version runs inside every
    bench.py run, and the suite has to stay well inside the driver's time limit): every mode the bench times — tile sweeps
    (interpreted, run-time-compiled, relabelled), the 1e-12 modes (tile = 2, fused multiply-adds, dense fusion) and the other
    BASELINE circuits (QFT, Clifford+T, Grover) through run-time-compiled sweeps — against the oracle on closed sub-cubes, with
    a twin state that goes gate by gate through the literal kernel compared over ALL 2^n amplitudes after every step, and
    closed-form marginals while the state is a product state."""
    from oracle import window_parity as W

    ops0, vecs = W.product_state_ops(n, seed=n)
    c2s = circuits.c2_random_circuit(n, 16, seed=28, single_only=True)
    c2 = circuits.c2_random_circuit(n, 5 * 16, seed=29)
    with q.HipState(n) as st:
        st.init_basis(0)
        st.apply_ops(ops0)
        twin = W.Twin(st, lambda: q.HipState(n))
        guard = W.ProductGuard(n, vecs)
        assert guard.check(st) < 1e-12
        agg = W.check_circuit(st, n, c2s, O, gate_by_gate=True, seed=1, twin=twin)
        assert agg["bit_equal"] and agg["whole_vector_amplitudes_not_equal"] == 0 and agg["whole_vector_compares"] == 16, agg
        for op in c2s:
            guard.apply(op)
        assert guard.check(st) < 1e-11, guard.worst_rel

        def leg(ops, exact, max_len=64, **options):
            for k, v in options.items():
                st.set_option(k, v)
            r = W.check_circuit(st, n, ops, O, gate_by_gate=False, seed=len(ops), bases_per_step=2, twin=twin, max_len=max_len)
            for k in options:
                st.set_option(k, 0)
            assert r["skipped"] == 0 and r["gates"] == len(ops), (options, r)
            if exact:
                assert r["bit_equal"] and r["whole_vector_amplitudes_not_equal"] == 0, (options, r)
            else:
                assert r["max_abs_delta"] <= TOL64 and r["whole_vector_max_abs_delta"] <= TOL64, (options, r)
                twin.resync()
            return r

        leg(c2[:16], True, tile=1)
        leg(c2[16:32], True, tile=1, tile_jit=1)
        leg(c2[32:48], True, tile=1, tile_jit=1, tile_relabel=2)
        leg(c2[48:64], False, tile=2, tile_jit=1)
        leg(c2[64:80], False, tile=2, tile_jit=1, tile_fma=1, tile_relabel=1)
        leg(circuits.c2_random_circuit(n, 16, seed=31), False, fuse=5)
        r = leg(circuits.c3_qft(n)[:120], True, max_len=160, tile=1, tile_jit=1)  # the first 5 H with all their controlled phases
        assert r["steps"] <= 2  # chunks as large as the timed segments (5-6 H and their controlled phases each)
        leg(circuits.c4_clifford_t(n, 48, seed=32), True, tile=1, tile_jit=1)
        leg(circuits.c5_grover_iteration(n)[:70], True, max_len=96, tile=1, tile_jit=1)  # X / H walls and the 27-control Z
        # r4: wide tiles (13-bit register-resident tile, seven free positions per sweep): IEEE-equal in circuit order, also
        # with the qubits relabelled; the 1e-12 mode with commuting reorder
        # (bench.py's parity block checks every wide leg it times at n = 30 — Clifford+T, Grover and the relabelled 1e-12 mode too;
        # test_wide_tiles_… compares wide with narrow sweeps bit for bit at n = 18)
        # (r5: two wide legs here — the relabelled and the QFT ones are bench.py parity legs at n = 30 and test_wide_tiles_… cases)
        c2w = circuits.c2_random_circuit(n, 40, seed=33)
        leg(c2w, True, tile=1, tile_jit=1, tile_wide=1)
        leg(circuits.c2_random_circuit(n, 40, seed=35), False, tile=2, tile_jit=1, tile_wide=1, tile_fma=1, tile_merge=1)
        twin.close()
        assert abs(st.norm_sqr() - 1) < 1e-9


def _jit_info():
    import ctypes as C

    from rustqip_amd import _ffi

    k, ms = C.c_uint64(), C.c_double()
    assert _ffi.lib.qip_hip_jit_stats(C.byref(k), C.byref(ms)) == 0
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


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_parametrised_segments_variational_loop_compiles_once(O, dtype):
    """option tile_jit = 1: the segment's structure is code, its numbers are kernel data.  64 angles updated 8 times (f32: 4): the
    number of compiled kernels stays what the first pass made it; every pass is bit-identical to the interpreter sweeps and
    to tile_jit = 3 — numbers as literals — (same arithmetic per amplitude), and the f64 result equals the oracle's."""
    n = 16
    rng = np.random.default_rng(5)
    x = circuits.random_state(n, seed=3, dtype=dtype)
    compiled_after_first = None
    passes = 8 if dtype == np.complex128 else 4  # (r5: 20 before — every pass of the literal form is a set of fresh compilations)
    with q.HipState(n, dtype) as st, q.HipState(n, dtype) as ref:
        st.set_option("tile", 1)
        st.set_option("tile_jit", 1)
        ref.set_option("tile", 1)
        for it in range(passes):
            ops = _ansatz(n, rng.uniform(0.05, 3.0, (4, n)))  # 64 angles
            st.upload(x)
            st.apply_ops(ops)
            got = st.download()
            ref.upload(x)
            ref.apply_ops(ops)
            assert np.array_equal(got, ref.download()), it
            if it == 0:
                compiled_after_first = _jit_info()["compiled"]
                if dtype == np.complex128:
                    assert np.array_equal(got, O.apply_ops_in_place(n, ops, x.copy()))
                ref.set_option("tile_jit", 3)  # from here on the reference is the run-time-compiled form with literal numbers
        assert _jit_info()["compiled"] - compiled_after_first >= passes - 1  # the literal form compiled new kernels every pass ...
        st_only = _jit_info()["compiled"]
        ops = _ansatz(n, rng.uniform(0.05, 3.0, (4, n)))
        st.upload(x)
        st.apply_ops(ops)
        assert _jit_info()["compiled"] == st_only  # ... the parametrised form none after its first
        # as a captured program: the parameters travel through the arena; re-recording with new angles reuses the kernels
        st.upload(x)
        prog = st.compile_program(ops)
        prog.run()
        assert prog.is_graph and _jit_info()["compiled"] == st_only
        ref.upload(x)
        ref.apply_ops(ops)
        assert np.array_equal(st.download(), ref.download())
        prog.close()


def test_jit_cache_is_bounded_and_programs_survive_evictions():
    """global option jit_cache_cap: least recently used kernels are unloaded beyond the bound; a program recorded into a
    hipGraph notices that an eviction happened since and re-records (its kernels are compiled again) instead of launching
    an unloaded module."""
    n = 14
    x = circuits.random_state(n, seed=8)
    base = _jit_info()
    q.set_global_option("jit_cache_cap", 3)
    try:
        with q.HipState(n) as st, q.HipState(n) as ref:
            st.set_option("tile", 1)
            st.set_option("tile_jit", 1)
            first = circuits.c2_random_circuit(n, 40, seed=1)
            st.upload(x)
            prog = st.compile_program(first)
            prog.run()
            assert prog.is_graph
            ref.upload(x)
            ref.apply_ops(first)
            assert np.array_equal(st.download(), ref.download())
            for seed in range(2, 8):  # new sources push the program's kernels out
                st.apply_ops(circuits.c2_random_circuit(n, 40, seed=seed))
            info = _jit_info()
            assert info["resident"] <= 3 and info["evicted"] > base["evicted"] and info["cap"] == 3
            st.upload(x)
            prog.run()  # re-records: the evicted kernels are compiled again
            assert np.array_equal(st.download(), ref.download())
            assert _jit_info()["compiled"] > info["compiled"]
            prog.close()
    finally:
        q.set_global_option("jit_cache_cap", 512)


def test_two_threads_compile_and_run_segments_at_once():
    """'separate handles are independent': two host threads, each with its own state, both with tile_jit, hammering the
    process-wide run-time compiler (loader, cache, counters) at the same time — results stay bit-identical to the
    interpreter sweeps."""
    import threading

    n = 13
    errors = []

    def work(tid):
        try:
            rng = np.random.default_rng(100 + tid)
            x = circuits.random_state(n, seed=tid)
            with q.HipState(n) as st, q.HipState(n) as ref:
                st.set_option("tile", 1)
                st.set_option("tile_jit", 1 + 2 * (tid % 2))
                ref.set_option("tile", 1)
                for it in range(12):
                    ops = circuits.c2_random_circuit(n, 30, seed=int(rng.integers(0, 1 << 30)))
                    st.upload(x)
                    st.apply_ops(ops)
                    ref.upload(x)
                    ref.apply_ops(ops)
                    if not np.array_equal(st.download(), ref.download()):
                        errors.append((tid, it, "mismatch"))
        except Exception as exc:  # noqa: BLE001
            errors.append((tid, repr(exc)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_sparse_on_many_qubits_out_of_place_ell_kernel(O, dtype):
    """SparseMatrix on k >= 6 qubits with <= 4 entries per row (k_sparse_ell; the reference's bench shape is a 16-qubit
    sparse identity, state_bench.rs:380-393): rows in stored order folded from 0, repeated columns and ragged rows included,
    optionally controlled, targets in any order — bit-equal to the oracle; wider rows fall back to the literal kernel."""
    n = 18
    rng = np.random.default_rng(11)
    x = circuits.random_state(n, seed=5, dtype=dtype)

    def rand_rows(k, width, phases=True):
        rows = []
        for r in range(1 << k):
            w = int(rng.integers(1, width + 1))
            ent = []
            for _ in range(w):
                c = int(rng.integers(0, 1 << k))
                v = complex(np.exp(1j * rng.uniform(0, 6))) if phases else complex(rng.standard_normal(), rng.standard_normal())
                ent.append((c, v))
            rows.append(ent)
        return rows

    k16 = [[(r, 1.0)] for r in range(1 << 16)]                                   # the bench's identity
    perm = rng.permutation(1 << 8)
    perm_phase = [[(int(perm[r]), complex(np.exp(1j * 0.1 * r)))] for r in range(1 << 8)]  # a generalised permutation
    cases = [
        ("identity16", q.make_sparse_matrix_op(list(range(16)), k16)),
        ("identity16_scattered", q.make_sparse_matrix_op([int(v) for v in rng.permutation(n)[:16]], k16)),
        ("perm_phase8", q.make_sparse_matrix_op([17, 0, 9, 3, 12, 5, 1, 16], perm_phase)),
        ("two_per_row6", q.make_sparse_matrix_op([2, 17, 8, 0, 11, 5], rand_rows(6, 2))),
        ("four_per_row7_ragged", q.make_sparse_matrix_op([4, 1, 16, 9, 13, 0, 7], rand_rows(7, 4, phases=False))),
        ("controlled6", q.make_control_op([3, 17], q.make_sparse_matrix_op([0, 6, 10, 12, 15, 1], rand_rows(6, 3)))),
        ("five_per_row6_literal", q.make_sparse_matrix_op([2, 17, 8, 0, 11, 5], rand_rows(6, 5))),
        # r4, k_sparse_tile's corners (qubit q is index position n-1-q; the wave row is positions 0..4 and 11 for Complex<f64>,
        # 0..5 for Complex<f32>): seven positions outside the row (128 KiB of LDS in f64), three, the row's split position as an
        # op bit and as a control, position 5 as an op bit, every op bit but three inside the row, controls inside and outside
        ("tile_kh7", q.make_sparse_matrix_op([17, 0, 9, 3, 11, 5, 1, 2], perm_phase)),
        ("tile_kh3", q.make_sparse_matrix_op([17, 16, 15, 1, 9, 4], rand_rows(6, 2))),
        ("tile_split_position_op", q.make_sparse_matrix_op([6, 0, 12, 3, 9, 1], rand_rows(6, 4))),
        ("tile_split_position_ctl", q.make_control_op([6, 12], q.make_sparse_matrix_op([0, 2, 4, 8, 10, 17], rand_rows(6, 2)))),
        ("tile_ctl_in_row_and_out", q.make_control_op([15, 1, 13], q.make_sparse_matrix_op([0, 2, 4, 8, 10, 17, 16], rand_rows(7, 3)))),
        ("ell_only_two_outside", q.make_sparse_matrix_op([17, 16, 15, 14, 13, 0, 1], rand_rows(7, 2))),
    ]
    tile_cases = {"perm_phase8", "two_per_row6", "four_per_row7_ragged", "controlled6"} | {c[0] for c in cases if c[0].startswith("tile_")}
    with q.HipState(n, dtype) as st:
        st.set_option("profile", 1)
        for name, op in cases:
            st.upload(x)
            st.profile_reset()
            st.apply_op(op)
            got = st.download()
            want = O.apply_ops_in_place(n, [op], x.copy())
            assert np.array_equal(got, want), name
            prof = st.profile()
            assert ("k_gather_generic" in prof) == (name == "five_per_row6_literal"), (name, prof)
            assert ("k_sparse_tile" in prof) == (name in tile_cases), (name, prof)
            assert ("k_sparse_ell" in prof) == (name not in tile_cases and name != "five_per_row6_literal"), (name, prof)
        # the out-of-place gather on the same ops (global option sparse_tile = 0): the very same bits
        q.set_global_option("sparse_tile", 0)
        try:
            for name, op in cases:
                if name not in tile_cases:
                    continue
                st.upload(x)
                st.profile_reset()
                st.apply_op(op)
                assert np.array_equal(st.download(), O.apply_ops_in_place(n, [op], x.copy())), name
                assert "k_sparse_ell" in st.profile(), name
        finally:
            q.set_global_option("sparse_tile", 1)
    # the smallest state the tile form takes (6 + kh + outside controls + 2 positions), and one below it
    for nn in (11, 12, 13):
        xs = circuits.random_state(nn, seed=nn, dtype=dtype)
        op = q.make_control_op([0], q.make_sparse_matrix_op([1, 2, 3, nn - 1, nn - 2, 5], rand_rows(6, 2)))
        with q.HipState(nn, dtype) as st:
            st.upload(xs)
            st.apply_op(op)
            assert np.array_equal(st.download(), O.apply_ops_in_place(nn, [op], xs.copy())), nn


def test_soft_measure_map_sample_sweep_f64_and_f32(O):
    """How often does the device's sample -> outcome map differ from the reference's sequential scan
    (measurement_ops.rs:153-176: r -= |amp_i|^2 until r <= 0)?  10^4 samples each.
    f64: the device subtracts chunk sums (summed in another order) and replays only the crossing chunk sequentially, so a
    disagreement needs the sample within rounding of a chunk boundary: 0 of 10^4 here.
    f32: the reference subtracts 2^n single-precision numbers from a single-precision r one after the other — every
    subtraction rounds to ~6e-8 relative of r, so at n = 14 its own crossing point is already off by thousands of ulps and
    amplitudes below r * 6e-8 do not move r at all — while the device accumulates the chunk sums in double.  The two maps
    therefore agree only where the sample is far from a boundary on the f32 scale: the fraction that differs is measured and
    bounded here (it is the reference's rounding, not the device's), and the distributions agree (chi-square over outcomes)."""
    rng = np.random.default_rng(2024)
    samples = rng.uniform(0, 1, 10000)
    n = 14
    idx = [0, 5, 13]
    x = rand_state(n, 77)
    with q.HipState(n) as st:
        st.upload(x)
        got = np.array([st.soft_measure(idx, float(r)) for r in samples])
    want = np.array([O.soft_measure(n, idx, x, float(r)) for r in samples])
    assert int(np.count_nonzero(got != want)) == 0
    # r4: one launch (chunk sums + the last block's walk and replay; option soft_measure_one_pass, measured slower and off) against
    # the two-launch form (r4: two-level replay, coalesced segment sums): the same function of
    # the sample, also at the edges (r = 0, r = 1, r above the norm: never crossing -> outcome of index 0), on a state whose first
    # and last amplitudes are zero, at a size with 4096 chunks of 2^10 and at one with two chunks
    for nn, dt in ((22, np.complex128), (22, np.complex64), (11, np.complex128)):
        xs = rand_state(nn, 5 + nn, dt)
        xs[:3000 if nn > 11 else 5] = 0
        xs[-(2000 if nn > 11 else 3):] = 0
        xs /= np.sqrt(np.sum(np.abs(xs.astype(np.complex128)) ** 2)).astype(xs.real.dtype)
        ii = [0, nn // 2, nn - 1, 3]
        rs = [0.0, 1.5, 1e-300, 0.5, 0.99, 2.0] + [float(v) for v in rng.uniform(0, 1, 300)]  # (r = 1 exactly is decided by the summation order)
        with q.HipState(nn, dt) as st:
            st.upload(xs)
            two = [st.soft_measure(ii, r) for r in rs]
            q.set_global_option("soft_measure_one_pass", 1)
            try:
                one = [st.soft_measure(ii, r) for r in rs]
            finally:
                q.set_global_option("soft_measure_one_pass", 0)
        assert one == two, (nn, dt, [(r, a, b) for r, a, b in zip(rs, one, two) if a != b][:5])
        if dt == np.complex128:
            assert one[:150] == [O.soft_measure(nn, ii, xs, r) for r in rs[:150]], nn
    x32 = x.astype(np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.upload(x32)
        got32 = np.array([st.soft_measure(idx, float(r)) for r in samples])
    want32 = np.array([O.soft_measure(n, idx, x32, float(r)) for r in samples])
    differ = int(np.count_nonzero(got32 != want32))
    print(f"f32 soft_measure: {differ} of {len(samples)} samples map to another outcome than the reference's f32 scan")
    assert differ <= 100, differ  # ~1e-3 expected: samples within the f32 scan's accumulated rounding of an outcome boundary
    # the device's f32 outcomes are those of the exact (double) cumulative sums of the same f32 amplitudes
    p = np.abs(x32.astype(np.complex128)) ** 2
    cum = np.cumsum(p)
    exact_idx = np.minimum(np.searchsorted(cum, samples.astype(np.float32).astype(np.float64), side="left"), (1 << n) - 1)
    exact = np.array([sum(((int(i) >> (n - 1 - qq)) & 1) << b for b, qq in enumerate(idx)) for i in exact_idx])
    assert int(np.count_nonzero(got32 != exact)) <= 2
    probs = O.measure_probs(n, idx, x).astype(np.float64)
    for outcomes in (got32, want32):
        counts = np.bincount(outcomes, minlength=8).astype(np.float64)
        chi2 = float(np.sum((counts - len(samples) * probs) ** 2 / (len(samples) * probs)))
        assert chi2 < 40, chi2  # 7 degrees of freedom


def test_relabelled_layout_persists_across_apply_ops_calls(O):
    """option tile_relabel = 3: a circuit applied in chunks keeps the scheduler's qubit layout between the calls — fewer
    sweeps than paying the restoring permutation after every chunk — and whatever needs the caller's order (download,
    measurement, a gate-by-gate call, a second state's comparison) restores it first.  Only moves differ: bit-identical
    to the plain tile sweeps and to the oracle."""
    n = 18
    ops = circuits.c2_random_circuit(n, 240, seed=41)
    x = circuits.random_state(n, seed=4)
    want = O.apply_ops_in_place(n, ops, x.copy())

    def run(relabel, chunks):
        with q.HipState(n) as st:
            st.set_option("tile", 1)
            st.set_option("tile_relabel", relabel)
            st.set_option("profile", 1)
            st.upload(x)
            st.profile_reset()
            step = len(ops) // chunks
            for c in range(chunks):
                st.apply_ops(ops[c * step:(c + 1) * step if c + 1 < chunks else len(ops)])
            prof = st.profile()
            sweeps = sum(v["launches"] for v in prof.values())
            perms = prof.get("k_permute_bits", {}).get("launches", 0)
            got = st.download()
            return got, sweeps, perms

    plain, sweeps_plain, _ = run(0, 6)
    each, sweeps_each, perms_each = run(2, 6)     # relabel, restore after every chunk
    kept, sweeps_kept, perms_kept = run(3, 6)     # relabel, layout kept between the chunks
    assert np.array_equal(plain, want) and np.array_equal(each, want) and np.array_equal(kept, want)
    assert perms_kept == 0 and perms_each >= 4
    assert sweeps_kept < sweeps_each and sweeps_kept <= sweeps_plain, (sweeps_plain, sweeps_each, sweeps_kept)
    # everything that addresses amplitudes sees the caller's order
    with q.HipState(n) as st, q.HipState(n) as ref:
        st.set_option("tile", 1)
        st.set_option("tile_relabel", 3)
        st.upload(x)
        ref.upload(x)
        st.apply_ops(ops[:80])
        ref.apply_ops(ops[:80])
        assert np.array_equal(st.measure_probs([0, 7, n - 1]), ref.measure_probs([0, 7, n - 1]))  # settles
        st.apply_ops(ops[80:160])
        ref.apply_ops(ops[80:160])
        assert st.max_abs_diff(ref) == (0.0, 0)
        # ... from either side, and a copy of a relabelled state is a copy in the caller's order
        st.set_option("profile", 0)
        ref.set_option("tile", 1)
        ref.set_option("tile_relabel", 3)
        st.apply_ops(ops[:40])
        ref.apply_ops(ops[:40])          # both relabelled now (their layouts are the same plan's, but nothing relies on that)
        with q.HipState(n) as third:
            third.copy_from(ref)
            assert st.max_abs_diff(third) == (0.0, 0) and third.max_abs_diff(ref) == (0.0, 0)
        ref.set_option("tile", 0)
        ref.set_option("tile_relabel", 0)
        st.upload(x)
        ref.upload(x)
        st.apply_ops(ops[:160])
        ref.apply_ops(ops[:160])
        st.apply_ops(ops[160:200])
        st.apply_op(ops[200])  # a single op: gate-by-gate entry point
        ref.apply_ops(ops[160:201])
        assert np.array_equal(st.download(1000, 4096), ref.download(1000, 4096))
        st.apply_ops(ops[201:])
        ref.apply_ops(ops[201:])
        prog = st.compile_program(ops[:40])  # a capture starts and ends in the caller's order
        prog.run()
        ref.apply_ops(ops[:40])
        assert np.array_equal(st.download(), ref.download())
        prog.close()
        st.apply_ops(ops[40:120])
        st.init_basis(3)  # overwrites: no restoring sweep needed, and none left pending
        e = np.zeros(1 << n, dtype=np.complex128)
        e[3] = 1
        assert np.array_equal(st.download(), e)


@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-12), (np.complex64, 2e-5)])
def test_tile2_merged_diagonal_runs_and_fused_multiply_adds(O, dtype, tol):
    """options tile_merge / tile_fma (tile = 2, run-time-compiled; 1e-12 bar): runs of diagonal gates applied as products of
    their factors — QFT's controlled phases, multi-controlled phases with controls on lane, register and outside-the-tile bits,
    Rz layers — against the oracle, with and without contraction; tile = 1 ignores both options (stays bit-identical)."""
    n = 17
    rng = np.random.default_rng(3)
    diag_heavy = []
    for _ in range(120):
        perm = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 6))
        if kind == 0:
            diag_heavy.append(q.make_matrix_op([perm[0]], circuits.H))
        elif kind == 1:
            diag_heavy.append(q.make_matrix_op([perm[0]], circuits.rz(float(rng.uniform(0, 3)))))
        elif kind == 2:
            diag_heavy.append(q.make_control_op(perm[:1], q.make_matrix_op([perm[1]], [1, 0, 0, cmath.rect(1, float(rng.uniform(0, 3)))])))
        elif kind == 3:
            diag_heavy.append(q.make_control_op(perm[:3], q.make_matrix_op([perm[3]], [1, 0, 0, cmath.rect(1, float(rng.uniform(0, 3)))])))
        elif kind == 4:
            diag_heavy.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], circuits.rz(float(rng.uniform(0, 3))))))
        else:
            diag_heavy.append(q.make_matrix_op([perm[0]], circuits.T))
    x = circuits.random_state(n, seed=6, dtype=dtype)
    for name, ops in (("qft", circuits.c3_qft(n)), ("diag_heavy", diag_heavy), ("c4", circuits.c4_clifford_t(n, 120, seed=2))):
        want = O.apply_ops_in_place(n, ops, x.copy())
        for merge, fma in ((1, 0), (1, 1), (0, 1)):
            with q.HipState(n, dtype) as st:
                for k, v in (("tile", 2), ("tile_jit", 1), ("tile_merge", merge), ("tile_fma", fma)):
                    st.set_option(k, v)
                st.upload(x)
                st.apply_ops(ops)
                err = float(np.max(np.abs(st.download() - want)))
                assert err <= tol, (name, merge, fma, err)
                assert abs(st.norm_sqr() - 1) <= (1e-12 if dtype == np.complex128 else 1e-4)
        if dtype == np.complex128:
            with q.HipState(n) as st:
                for k, v in (("tile", 1), ("tile_jit", 1), ("tile_merge", 1), ("tile_fma", 1)):
                    st.set_option(k, v)
                st.upload(x)
                st.apply_ops(ops)
                assert np.array_equal(st.download(), want), name


@pytest.mark.parametrize("dtype,tol", [(np.complex128, TOL64), (np.complex64, TOL32)])
def test_dense4_on_the_matrix_cores_through_an_lds_tile(O, dtype, tol):
    """k_gate_k4_tile_mfma: the matrix-core form of a dense 4-qubit gate with its operands staged through the one-op sweeps'
    tile (whole rows on both global sides).  Every placement of the targets — inside the rows, above them, mixed, adjacent to
    the padding positions, with controls above the rows — against the oracle (fma chains: 1e-12 / 1e-5), against the
    direct-from-HBM kernel (same fragments, same chains: identical), and the fall-back for a control inside a row."""
    n = 19
    rng = np.random.default_rng(19)
    x = circuits.random_state(n, seed=2, dtype=dtype)
    u = rand_unitary(4, rng)
    perm01 = np.eye(16)[rng.permutation(16)]  # a 0/1 permutation matrix stays exact on the matrix cores
    cases = []
    for targets in ([18, 17, 16, 15], [18, 17, 0, 1], [0, 7, 18, 9], [12, 18, 17, 3], [5, 6, 7, 8], [0, 1, 2, 3], [18, 16, 14, 12]):
        cases.append((f"targets {targets}", q.make_matrix_op(targets, u.ravel()), targets))
    cases.append(("controlled, controls above the rows", q.make_control_op([2, 9], q.make_matrix_op([18, 17, 0, 5], u.ravel())), None))
    cases.append(("controlled, a control inside a row (direct kernel)", q.make_control_op([16], q.make_matrix_op([18, 17, 0, 5], u.ravel())), None))
    with q.HipState(n, dtype) as st, q.HipState(n, dtype) as direct:
        st.set_option("mfma", 2)      # the matrix-core form for every placement (default: when two or more targets are low)
        direct.set_option("mfma", 2)
        for name, op, _ in cases:
            st.upload(x)
            st.apply_op(op)
            got = st.download()
            want = O.apply_ops_in_place(n, [op], x.copy())
            assert float(np.max(np.abs(got - want))) <= tol, name
            q.set_global_option("k4_direct", 1)
            try:
                direct.upload(x)
                direct.apply_op(op)
            finally:
                q.set_global_option("k4_direct", 0)
            assert np.array_equal(got, direct.download()), name
        st.upload(x)
        st.apply_op(q.make_matrix_op([18, 17, 0, 1], perm01.ravel()))
        assert np.array_equal(st.download(), O.apply_ops_in_place(n, [q.make_matrix_op([18, 17, 0, 1], perm01.ravel())], x.copy()))


@pytest.mark.parametrize("row_split", [11, 5])
def test_one_op_tile_sweeps_controlled_dense_and_both_row_shapes(O, row_split):
    """r4: (a) a CONTROLLED dense k = 2 / 3 gate runs as a one-op tile sweep too — controls above the rows come off the grid
    (half / quarter sweeps), controls inside a row or a 128-byte line are lane predicates — with the unfused register fold:
    bit-equal to the oracle (ControlledOpIterator, qubit_iterators.rs:124-171); (b) the tile's rows in both shapes: split
    (two 512-byte halves 32 KiB apart, positions {0..4, 11}; position 5 is then an ordinary high position) and contiguous."""
    n = 20
    rng = np.random.default_rng(7)
    x = rand_state(n, 3)
    u2, u3 = rand_unitary(2, rng), rand_unitary(3, rng)
    q.set_global_option("tile_row_split", row_split)
    try:
        cases = []
        # position p <-> qubit n-1-p.  Targets / controls on: a line bit (0..2), a row bit (3, 4), position 5, 11, 12, high ones
        P = lambda *pos: [n - 1 - p for p in pos]  # noqa: E731
        for tg in (P(0, 1), P(4, 5), P(5, 11), P(11, 12), P(2, 17), P(19, 18), P(5, 6)):
            cases.append(("dense2", q.make_matrix_op(tg, u2.ravel())))
            for ct in (P(3), P(9), P(13, 1), P(10, 14)):
                if not set(ct) & set(tg):
                    cases.append(("cdense2", q.make_control_op(ct, q.make_matrix_op(tg, u2.ravel()))))
        for tg in (P(0, 1, 2), P(5, 11, 4), P(19, 11, 5), P(16, 17, 18), P(3, 12, 15)):
            cases.append(("dense3", q.make_matrix_op(tg, u3.ravel())))
            for ct in (P(6), P(7, 13), P(1) if 1 not in [n - 1 - t for t in tg] else P(8)):
                if not set(ct) & set(tg):
                    cases.append(("cdense3", q.make_control_op(ct, q.make_matrix_op(tg, u3.ravel()))))
        for tq in (5, 11, 12, 19, 6):  # single-qubit gates above the rows, swaps with a bit inside a row
            cases.append(("h", q.make_matrix_op(P(tq), circuits.H)))
            cases.append(("swap", q.make_swap_op(P(tq), P(2))))
        with q.HipState(n) as st:
            st.set_option("profile", 1)
            for name, op in cases:
                st.upload(x)
                st.profile_reset()
                st.apply_op(op)
                got = st.download()
                want = oracle_apply(O, n, op, x)
                assert np.array_equal(got, want), (name, op.indices, row_split)
                if name in ("cdense2", "cdense3", "dense2", "dense3"):
                    assert "k_tile_passes" in st.profile(), (name, op.indices, st.profile())
            # the profile credits a controlled sweep with its algorithmic bytes (half the vector per control)
            st.profile_reset()
            st.apply_op(q.make_control_op(P(15, 16), q.make_matrix_op(P(0, 9), u2.ravel())))
            pr = st.profile()["k_tile_passes"]
            assert pr["algorithmic_bytes"] == 32.0 * 2 ** (n - 2), pr
        xf = rand_state(n, 4, np.complex64)
        for name, op in cases[::5]:
            got = hip_apply(n, op, xf)
            want = oracle_apply(O, n, op, xf)
            assert np.max(np.abs(got - want)) <= TOL32, (name, op.indices)
    finally:
        q.set_global_option("tile_row_split", 11)


@pytest.mark.slow
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_wide_tiles_match_the_narrow_sweeps_and_the_oracle(O, dtype):
    """r4, option tile_wide: run-time-compiled segments over a 13-bit tile held in registers (32 amplitudes per lane, seven free
    positions per sweep, LDS as a transposition buffer).  Same helpers and gate order as the 11-bit sweeps: tile = 1 is
    IEEE-equal to them (and so to the gate-by-gate path and the oracle); tile = 2 and the relabelled plans to the 1e-12 bar.
    Circuits with every item kind: single-qubit gates, CNOTs, multi-controlled gates with controls in rows / above / on register
    bits, diagonal gates on every kind of bit, swaps, dense 2- and 3-qubit gates."""
    n = 18
    f64 = dtype == np.complex128
    tol = TOL64 if f64 else TOL32
    rng = np.random.default_rng(18)
    u2, u3 = rand_unitary(2, rng), rand_unitary(3, rng)
    x = rand_state(n, 5, dtype)
    extra = []
    for _ in range(10):
        qs = [int(v) for v in rng.permutation(n)]
        extra += [q.make_matrix_op(qs[:2], u2.ravel()), q.make_control_op(qs[2:4], q.make_matrix_op([qs[4]], circuits.H)),
                  q.make_matrix_op(qs[5:8], u3.ravel()), q.make_swap_op([qs[8]], [qs[9]]),
                  q.make_control_op([qs[10]], q.make_matrix_op([qs[11]], [1, 0, 0, cmath.rect(1, 0.3)])),
                  q.make_control_op([qs[12]], q.make_matrix_op(qs[13:15], u2.ravel())), q.make_matrix_op([qs[15]], circuits.rz(0.7))]
    cases = {"c2": circuits.h_layer(n) + circuits.c2_random_circuit(n, 200, seed=28),
             "c4": circuits.c4_clifford_t(n, 160, seed=32),
             "qft": circuits.c3_qft(n),
             "grover_k3": circuits.c5_grover_iteration(n, dense_k3=True),
             "mixed_items": circuits.c2_random_circuit(n, 40, seed=3) + extra}
    if not f64:  # (the f32 generator differs from the f64 one in the element type only: three circuits keep the suite's time down)
        cases = {k: cases[k] for k in ("c2", "qft", "mixed_items")}
    for name, ops in cases.items():
        want = O.apply_ops_in_place(n, ops, x.copy())
        for tile, relabel in ((1, 0), (1, 2), (2, 0), (2, 1)):
            res = {}
            for wide in (0, 1):
                with q.HipState(n, dtype) as st:
                    for k, v in (("tile", tile), ("tile_jit", 1), ("tile_relabel", relabel), ("tile_wide", wide), ("profile", 1)):
                        st.set_option(k, v)
                    st.upload(x)
                    st.apply_ops(ops)
                    res[wide] = (st.download(), sum(v["launches"] for v in st.profile().values()))
            assert float(np.max(np.abs(res[1][0] - want))) <= tol, (name, tile, relabel)
            if tile == 1 and relabel == 0 and f64 and name != "grover_k3":
                assert np.array_equal(res[1][0], res[0][0]), (name, "wide and narrow circuit-order sweeps differ")
            assert res[1][1] <= res[0][1], (name, tile, relabel, res[0][1], res[1][1])  # never more sweeps than the narrow plan
        if name == "c4" and f64:  # global option tile_wide_pin (register pins after block-uniform branches: no semantics; default on): the very same bits without
            q.set_global_option("tile_wide_pin", 0)
            try:
                with q.HipState(n, dtype) as st:
                    for k, v in (("tile", 1), ("tile_jit", 1), ("tile_wide", 1)):
                        st.set_option(k, v)
                    st.upload(x)
                    st.apply_ops(ops)
                    pinned = st.download()
            finally:
                q.set_global_option("tile_wide_pin", 1)
            with q.HipState(n, dtype) as st:
                for k, v in (("tile", 1), ("tile_jit", 1), ("tile_wide", 1)):
                    st.set_option(k, v)
                st.upload(x)
                st.apply_ops(ops)
                assert np.array_equal(pinned, st.download())
        if name == "grover_k3":
            # global option tile_wide_dense3_inline (a generator branch that had never run on a GPU): dense 3-qubit gates
            # written out group by group are the SAME fold as pass_dense3w (same products, same order) -> the very same bits
            both = {}
            for inline in (0, 1):
                q.set_global_option("tile_wide_dense3_inline", inline)
                try:
                    with q.HipState(n, dtype) as st:
                        for k, v in (("tile", 1), ("tile_jit", 1), ("tile_wide", 1)):
                            st.set_option(k, v)
                        st.upload(x)
                        st.apply_ops(ops)
                        both[inline] = st.download()
                finally:
                    q.set_global_option("tile_wide_dense3_inline", WIDE_DENSE3_INLINE_DEFAULT)
            assert np.array_equal(both[0], both[1]), "dense 3-qubit gates written out group by group differ from pass_dense3w"
            assert float(np.max(np.abs(both[1] - want))) <= tol
        if name in ("qft", "c4"):  # merged runs of diagonal gates + fused multiply-adds in the wide generator (1e-12 mode)
            with q.HipState(n, dtype) as st:
                for k, v in (("tile", 2), ("tile_jit", 1), ("tile_wide", 1), ("tile_fma", 1), ("tile_merge", 1)):
                    st.set_option(k, v)
                st.upload(x)
                st.apply_ops(ops)
                assert float(np.max(np.abs(st.download() - want))) <= tol, (name, "merged")
    # the smallest states that take wide tiles at all (one and two positions above the 13-bit tile), and n = 13 (narrow sweeps: no room)
    for nn in (13, 14, 15):
        xs = rand_state(nn, nn, dtype)
        ops = circuits.h_layer(nn) + circuits.c2_random_circuit(nn, 120, seed=nn) + circuits.c3_qft(nn)[:60]
        want = O.apply_ops_in_place(nn, ops, xs.copy())
        for tile in (1, 2):
            with q.HipState(nn, dtype) as st:
                for k, v in (("tile", tile), ("tile_jit", 1), ("tile_wide", 1)):
                    st.set_option(k, v)
                st.upload(xs)
                st.apply_ops(ops)
                got = st.download()
            assert float(np.max(np.abs(got - want))) <= tol, (nn, tile)
