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
