"""every op kind and kernel class against the CPU oracle.
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture
def line_bits(request):
    if request.param != 3:
        needs_tuning()  # (3 is what the product build fixes; the other thresholds are a tuning build's)
    if tuning():
        q.set_global_option("line_bits", request.param)
    yield request.param
    if tuning():
        q.set_global_option("line_bits", 3)


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


@pytest.mark.parametrize("name", sorted(GATES_1Q))
@pytest.mark.parametrize("n", [1, 2, 5, 11])
def test_single_qubit_all_targets(O, name, n):
    for target in range(n):
        op = q.make_matrix_op([target], GATES_1Q[name])
        x = rand_state(n, 100 + target)
        want = oracle_apply(O, n, op, x)
        for opts in variants({}, {"lowbit_shuffle": 0}, {"force_generic": 1}):
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
            for opts in variants({}, {"lowbit_shuffle": 0}, {"force_generic": 1}):
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
                for opts in variants({}, {"lowbit_shuffle": 0}):
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
    if k <= 8:  # r6: a matrix without an imaginary part goes through the two-product form (k = 6..8), a complex one through three
        orth = np.linalg.qr(rng.standard_normal((1 << k, 1 << k)))[0]
        for idx in (list(range(n - k, n)), [int(v) for v in rng.permutation(n)[:k]]):
            op = q.make_matrix_op(idx, orth.ravel())
            assert np.max(np.abs(hip_apply(n, op, x) - oracle_apply(O, n, op, x))) <= TOL64, (k, idx)
            opf = q.make_matrix_op(idx, orth.astype(np.complex64).ravel())
            xf32 = rand_state(n, 6, np.complex64)
            assert np.max(np.abs(hip_apply(n, opf, xf32) - oracle_apply(O, n, opf, xf32))) <= TOL32, (k, idx)
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
    op = q.make_matrix_op(list(range(8)), u.ravel())
    check(O, 8, op, bitwise=False, paths=("fast",))  # (r6: k_dense_small — partial sums of 16 columns: the 1e-12 bar of dense k >= 3)
    check(O, 8, op, paths=("generic",))               # the literal fold stays bit-equal


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
        if tuning():
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
            if tuning():  # (operands straight from HBM: the same fma chain, a tuning build's option)
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


@pytest.mark.parametrize("dtype,tol", [(np.complex128, TOL64), (np.complex64, TOL32)])
def test_dense5_on_the_matrix_cores_through_an_lds_tile(O, dtype, tol):
    """k_gate_tile_mfma<K = 5> (r6): a dense 5-qubit gate of a Complex<f64> state of >= 23 qubits with its operands staged through
    the one-op sweeps' tile — one item of 16 groups per wave, a block walking 8+ consecutive tiles with the next tile's rows in
    flight.  Every placement of the targets (inside the rows, above them, mixed, on the split position 11 and on position 5, all
    five above the rows), controls above the rows (one control: 8 tiles per block at n = 24), the fall-backs to the direct
    kernel (a control inside a row, smaller states, Complex<f32>) — against the oracle (fma chains: 1e-12 / 1e-5), against the
    direct-from-HBM kernel (same fragments: identical) in a tuning build; a 0/1 matrix stays exact."""
    rng = np.random.default_rng(23)
    u = rand_unitary(5, rng)
    perm01 = np.eye(32)[rng.permutation(32)]
    for n in (23, 24, 19):
        x = circuits.random_state(n, seed=n, dtype=dtype)
        P = lambda *pos: [n - 1 - p for p in pos]  # noqa: E731  (position p <-> qubit n-1-p)
        cases = []
        if n != 24:
            for tg in (P(0, 1, 2, 3, 4), P(11, 5, 0, 1, 2), P(n - 1, n - 2, n - 3, n - 4, n - 5), P(n - 1, 3, 11, 7, 5), P(6, 7, 8, 9, 10), P(12, 0, n - 1, 5, 13), P(4, 11, 12, 13, 14)):
                cases.append((f"targets at positions {[n - 1 - t for t in tg]}", q.make_matrix_op(tg, u.ravel())))
            cases.append(("0/1 permutation matrix", q.make_matrix_op(P(n - 1, 1, 5, 11, 8), perm01.ravel())))
            # a matrix without an imaginary part takes the two-product form (P = G only): a real orthogonal matrix, H on five qubits
            cases.append(("real orthogonal matrix", q.make_matrix_op(P(n - 2, 0, 11, 6, 14), np.linalg.qr(rng.standard_normal((32, 32)))[0].ravel())))
            h5 = np.array([[1.0]])
            for _ in range(5):
                h5 = np.kron(h5, np.array([[1.0, 1.0], [1.0, -1.0]]) / math.sqrt(2.0))
            cases.append(("H on five qubits", q.make_matrix_op(P(n - 1, n - 3, 4, 5, 12), h5.ravel())))
        if n != 23:
            cases.append(("controlled, a control above the rows", q.make_control_op(P(15), q.make_matrix_op(P(n - 1, 0, 5, 11, 13), u.ravel()))))
            cases.append(("controlled, a control inside a row (direct kernel)", q.make_control_op(P(2), q.make_matrix_op(P(n - 1, 0, 5, 11, 13), u.ravel()))))
        with q.HipState(n, dtype) as st, q.HipState(n, dtype) as direct:
            for name, op in cases:
                st.upload(x)
                st.apply_op(op)
                got = st.download()
                want = O.apply_ops_in_place(n, [op], x.copy())
                if name.startswith("0/1"):
                    assert np.array_equal(got, want), (n, name)
                else:
                    assert float(np.max(np.abs(got - want))) <= tol, (n, name)
                if tuning():
                    q.set_global_option("k4_direct", 1)
                    try:
                        direct.upload(x)
                        direct.apply_op(op)
                    finally:
                        q.set_global_option("k4_direct", 0)
                    assert np.array_equal(got, direct.download()), (n, name)


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
    if row_split != 11:
        needs_tuning()  # (contiguous rows: rounds 1-3's layout, a tuning build's option)
    if tuning():
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
        if tuning():
            q.set_global_option("tile_row_split", 11)


def test_dense_gate_on_a_state_too_small_for_the_matrix_cores(O):
    """r6: dense k = 5..10 with fewer than 16 groups (n < k + controls + 4) — the reference's own bench shape, state_bench.rs:118-139
    (n = 8, one dense 8-qubit gate) — runs through k_dense_small (16 rows of a group per block, partial sums of S/16 columns) instead
    of the literal kernel: the 1e-12 bar of dense k >= 3 gates; force_generic keeps the literal fold (bit-equal)."""
    rng = np.random.default_rng(8)
    for k in (5, 6, 8, 10):
        for extra, nc in ((0, 0), (1, 0), (3, 0), (2, 1), (3, 2)):
            n = k + extra
            perm = [int(v) for v in rng.permutation(n)]
            op = q.make_matrix_op(perm[:k], rand_unitary(k, rng).ravel())
            if nc:
                op = q.make_control_op(perm[k:k + nc], op)
            for dtype, tol in ((np.complex128, TOL64), (np.complex64, TOL32)):
                x = rand_state(n, 3 * k + extra, dtype)
                want = oracle_apply(O, n, op, x)
                with q.HipState(n, dtype) as st:
                    st.set_option("profile", 1)
                    st.upload(x)
                    st.apply_op(op)
                    got = st.download()
                    prof = st.profile()
                assert float(np.max(np.abs(got - want))) <= tol, (k, n, nc, dtype)
                assert "k_dense_small" in prof and "k_gather_generic" not in prof, (k, n, nc, prof)
                if dtype == np.complex128:
                    assert np.array_equal(hip_apply(n, op, x, force_generic=1), want), (k, n, nc)
    # zero rows / zero columns and exact 0 / 1 entries survive (a permutation matrix stays exact: every sum has one non-zero term)
    k, n = 8, 9
    pm = np.zeros((256, 256))
    pm[np.arange(256), rng.permutation(256)] = 1.0
    op = q.make_matrix_op([8, 1, 2, 3, 4, 5, 6, 7], pm.ravel())
    x = rand_state(n, 5)
    assert np.array_equal(hip_apply(n, op, x), oracle_apply(O, n, op, x))
