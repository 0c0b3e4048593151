"""The full-size window checker (oracle/window_parity.py) itself, on the CPU: a numpy-backed stand-in for the device
state applies gates through the oracle on the FULL vector; the checker must agree with it bit for bit, and must
notice an index-mapping error that a norm check on a uniform state cannot see."""
import numpy as np
import pytest

import rustqip_amd as q
from oracle import qip_oracle as O
from oracle import window_parity as W
from rustqip_amd import circuits


class CpuState:
    def __init__(self, n, x, corrupt=None):
        self.n, self.x, self.corrupt = n, x.copy(), corrupt

    def download(self, offset=0, length=None):
        length = self.x.size - offset if length is None else length
        return self.x[offset:offset + length].copy()

    def apply_ops(self, ops):
        self.x = O.apply_ops_in_place(self.n, list(ops), self.x)
        if self.corrupt is not None:
            self.corrupt(self.x)


def _mixed_ops(n):
    rng = np.random.default_rng(3)
    u3 = np.linalg.qr(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))[0]
    return circuits.c2_random_circuit(n, 24, seed=5) + [
        q.make_swap_op([0, 3], [n - 1, n - 2]),
        q.make_control_op([1, n - 1], q.make_matrix_op([2], circuits.H)),
        q.make_matrix_op([0, n // 2, n - 1], u3.ravel()),
        q.make_sparse_matrix_op([n - 2, 1], [[(1, 0.5j)], [(0, 1), (3, 2)], [(2, -1)], [(0, 0.25)]]),
    ]


@pytest.mark.parametrize("w_max", [5, 8, 16])
def test_checker_agrees_with_full_vector_oracle(w_max):
    n = 14
    st = CpuState(n, circuits.random_state(n, 1))
    for op in _mixed_ops(n):
        r = W.check_ops(st, n, [op], O, w_max=w_max)
        assert r is not None and r["bit_equal"] and r["max_abs_delta"] == 0.0 and r["rows"] >= 4 << min(w_max, n)
        assert r["row_calls"] > 0


def test_checker_in_chunks_and_whole_circuit():
    n = 18
    ops = circuits.c2_random_circuit(n, 48, seed=28)
    st = CpuState(n, circuits.random_state(n, 2))
    agg = W.check_circuit(st, n, ops, O, gate_by_gate=False)
    assert agg["gates"] == 48 and agg["bit_equal"] and agg["skipped"] == 0 and agg["steps"] < 48
    ref = O.apply_ops_in_place(n, ops, circuits.random_state(n, 2))
    assert np.array_equal(st.x, ref)


def test_checker_sees_an_index_mapping_error():
    """exchange two amplitudes at the top of the index space after every gate: norm-preserving, invisible on a
    uniform state, caught by the window compare (the top of the index space is always one of the bases)"""
    n = 16

    def corrupt(x):
        x[[-1, -2]] = x[[-2, -1]]

    ops0, vecs = W.product_state_ops(n, 7)
    x = np.zeros(1 << n, dtype=np.complex128)
    x[0] = 1
    x = O.apply_ops_in_place(n, ops0, x)
    assert np.allclose(x[1000:1512], W.product_state_window(n, vecs, 1000, 512), atol=1e-14)
    assert len(np.unique(np.round(np.abs(x[:4096]), 14))) > 3000  # moduli are pairwise distinct, not uniform
    st = CpuState(n, x, corrupt)
    r = W.check_ops(st, n, [q.make_matrix_op([3], circuits.H)], O)
    assert r["max_abs_delta"] > 1e-6 and not r["bit_equal"]


def test_ops_touching_too_many_high_bits_are_skipped_not_misjudged():
    n = 30
    op = q.make_control_op(list(range(n - 1)), q.make_matrix_op([n - 1], circuits.Z))
    cube = W.SubCube(n, [n - 1 - i for i in op.indices])
    assert not cube.ok
    cube = W.SubCube(n, [29, 3, 17])
    assert cube.ok and cube.w == 16 and cube.high == [17, 29] and cube.m == 18
    assert cube.offsets((1 << n) - 1) == [((1 << n) - 1) & ~cube.vmask | h for h in (0, 1 << 17, 1 << 29, (1 << 17) | (1 << 29))]


class CpuTwinState(CpuState):
    """CpuState plus the two-state calls the whole-vector guard uses (HipState.copy_from / max_abs_diff / set_option)"""

    def set_option(self, key, value):
        pass

    def copy_from(self, other):
        self.x = other.x.copy()

    def max_abs_diff(self, other):
        d = np.abs(self.x - other.x)
        return float(d.max()), int(np.count_nonzero(~((self.x.real == other.x.real) & (self.x.imag == other.x.imag))))

    def close(self):
        pass


def test_tested_only_bits_may_lie_outside_the_cube_qft_and_many_controls():
    """Controls and diagonal targets outside the sub-cube are resolved against the cube's base (the op is the identity
    there, or the same op without that control / with the selected diagonal block): whole QFT chunks and a Z with n - 1
    controls are checked — not skipped — and agree bit for bit with the oracle on the full vector."""
    n = 20
    ops = circuits.c3_qft(n) + [q.make_control_op(list(range(n - 1)), q.make_matrix_op([n - 1], circuits.Z)),
                                q.make_control_op([0, n - 1], q.make_matrix_op([7, 3], np.diag([1, 1j, -1, np.exp(0.3j)]).ravel())),
                                q.make_matrix_op([1, n - 2], np.diag(np.exp(1j * np.arange(4))).ravel())]
    st = CpuState(n, circuits.random_state(n, 4))
    agg = W.check_circuit(st, n, ops, O, gate_by_gate=False, bases_per_step=6)
    assert agg["skipped"] == 0 and agg["gates"] == len(ops) and agg["bit_equal"] and agg["max_abs_delta"] == 0.0
    # QFT's chunks hold several H each (only the H targets must be closed over): far fewer steps than ops
    assert agg["steps"] <= len(ops) // 8
    assert np.array_equal(st.x, O.apply_ops_in_place(n, ops, circuits.random_state(n, 4)))
    # gate by gate with small cubes: every controlled phase has its control or its target outside the cube somewhere
    st = CpuState(n, circuits.random_state(n, 5))
    for op in circuits.c3_qft(n)[:60]:
        r = W.check_ops(st, n, [op], O, w_max=6, bases=W.default_bases(n, 3, 8))
        assert r is not None and r["bit_equal"]


def test_localize_at_matches_the_full_op_on_every_cube():
    """exhaustive on a small register: for every base, the localized op applied to the cube equals the full op's rows"""
    n = 9
    rng = np.random.default_rng(2)
    d8 = np.diag(np.exp(1j * rng.uniform(0, 6, 8)))
    ops = [q.make_control_op([0, 8], q.make_matrix_op([4], circuits.H)),                 # controls outside, dense target inside
           q.make_control_op([3], q.make_matrix_op([7, 1, 5], d8.ravel())),              # diagonal targets partly outside
           q.make_control_op([2], q.make_control_op([6], q.make_matrix_op([0], circuits.T))),  # nested control, target outside
           q.make_matrix_op([8], circuits.rz(0.7))]                                      # diagonal 1-qubit gate fully outside
    x = circuits.random_state(n, 9)
    for op in ops:
        full = O.apply_ops_in_place(n, [op], x.copy())
        cube = W.SubCube(n, W.exchange_positions(n, op), w_max=4)
        assert cube.ok
        seen_identity = seen_active = False
        for base in range(0, 1 << n, 1 << cube.w):
            if base & cube.vmask:
                continue
            sel = np.concatenate([np.arange(off, off + (1 << cube.w)) for off in cube.offsets(base)])
            lop = cube.localize_at(op, base)
            if lop is None:
                want = x[sel]
                seen_identity = True
            else:
                want = O.apply_ops_in_place(cube.m, [lop], x[sel].copy())
                seen_active = True
            assert np.array_equal(full[sel], want)
        assert seen_active


def test_twin_guard_sees_a_stray_write_outside_every_cube():
    """a corruption far from the compared windows: the sub-cubes stay clean, the whole-vector twin comparison does not"""
    n = 16
    hit = (1 << n) // 2 + 12345  # not in the bottom / top windows

    def corrupt(x):
        x[hit] *= 1.0000001

    x = circuits.random_state(n, 6)
    op = q.make_matrix_op([15], circuits.H)  # target bit 0: cubes are single windows of 2^w rows
    clean = CpuTwinState(n, x)
    tw = W.Twin(clean, lambda: CpuTwinState(n, x))
    r = W.check_ops(clean, n, [op], O, bases=[0, (1 << n) - 1], w_max=8, twin=tw)
    assert r["bit_equal"] and r["whole_vector"] == {"max_abs_delta": 0.0, "amplitudes_not_equal": 0}
    bad = CpuTwinState(n, x, corrupt)
    tw = W.Twin(bad, lambda: CpuTwinState(n, x))
    r = W.check_ops(bad, n, [op], O, bases=[0, (1 << n) - 1], w_max=8, twin=tw)
    assert r["bit_equal"]  # the windows cannot see it
    assert r["whole_vector"]["amplitudes_not_equal"] == 1 and r["whole_vector"]["max_abs_delta"] > 0


def test_product_guard_closed_form_marginals():
    """the closed-form marginals of the product state follow single-qubit gates exactly, and a misplaced amplitude shows"""
    n = 12
    ops0, vecs = W.product_state_ops(n, 3)
    x = np.zeros(1 << n, dtype=np.complex128)
    x[0] = 1
    x = O.apply_ops_in_place(n, ops0, x)

    class S:
        def __init__(self, x):
            self.x = x

        def measure_probs(self, idx):
            return O.measure_probs(n, list(idx), self.x)

    g = W.ProductGuard(n, vecs, k=5)
    assert sorted(set(t for s in g.sets for t in s)) == list(range(n))
    assert g.check(S(x)) < 1e-13
    for op in circuits.c2_random_circuit(n, 12, seed=4, single_only=True):
        x = O.apply_ops_in_place(n, [op], x)
        g.apply(op)
        assert g.check(S(x)) < 1e-12
    y = x.copy()
    y[[5, 4000]] = y[[4000, 5]]  # norm-preserving exchange of two amplitudes
    assert g.check(S(y)) > 1e-6


def test_localize_at_fuzz_random_ops_random_cubes():
    """300 random ops (dense with zero rows, block-diagonal, diagonal, sparse, swaps, nested controls; qubits anywhere) on a
    10-qubit register, cubes of random width: on EVERY cube the localized op reproduces the full op's rows bit for bit."""
    n = 10
    rng = np.random.default_rng(12345)
    x = circuits.random_state(n, 33)

    def rand_op():
        perm = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 8))
        if kind == 0:
            m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
            m[int(rng.integers(0, 2)), int(rng.integers(0, 2))] = 0  # a zero entry (zero-skipping)
            core, tg = q.make_matrix_op([perm[0]], m.ravel()), 1
        elif kind == 1:
            core, tg = q.make_matrix_op(perm[:2], np.diag(np.exp(1j * rng.uniform(0, 6, 4))).ravel()), 2
        elif kind == 2:
            u0, u1 = rng.standard_normal((2, 2)), rng.standard_normal((2, 2))
            m = np.zeros((4, 4), dtype=complex)
            m[:2, :2], m[2:, 2:] = u0, u1  # diagonal in its first target, dense in the second
            core, tg = q.make_matrix_op(perm[:2], m.ravel()), 2
        elif kind == 3:
            core, tg = q.make_swap_op(perm[:2], perm[2:4]), 4
        elif kind == 4:
            rows = [[(int(rng.integers(0, 4)), complex(rng.standard_normal(), rng.standard_normal()))
                     for _ in range(int(rng.integers(1, 3)))] for _ in range(4)]
            core, tg = q.make_sparse_matrix_op(perm[:2], rows), 2
        elif kind == 5:
            core, tg = q.make_matrix_op([perm[0]], [1, 0, 0, np.exp(0.37j)]), 1
        elif kind == 6:
            m = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
            core, tg = q.make_matrix_op(perm[:3], m.ravel()), 3
        else:
            core, tg = q.make_matrix_op(perm[:3], np.diag(np.exp(1j * rng.uniform(0, 6, 8))).ravel()), 3
        nc = int(rng.integers(0, 3))
        if nc and tg + nc <= n:
            return q.make_control_op(perm[tg:tg + nc], core)
        return core

    checked = 0
    for _ in range(300):
        op = rand_op()
        full = O.apply_ops_in_place(n, [op], x.copy())
        cube = W.SubCube(n, W.exchange_positions(n, op), w_max=int(rng.integers(2, 7)), m_max=n)
        if cube.w == 0:
            continue
        for base in range(0, 1 << n, 1 << cube.w):
            if base & cube.vmask:
                continue
            sel = np.concatenate([np.arange(off, off + (1 << cube.w)) for off in cube.offsets(base)])
            lop = cube.localize_at(op, base)
            want = x[sel] if lop is None else O.apply_ops_in_place(cube.m, [lop], x[sel].copy())
            assert np.array_equal(full[sel], want), (op.kind, op.indices, base)
            checked += 1
    assert checked > 5000
