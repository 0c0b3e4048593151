"""Complex<f32> states.
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


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
    if tuning():  # (the unpacked view of a Complex<f32> state: a tuning build's option)
        with q.HipState(n, np.complex64) as st:
            st.set_option("packed_f32", 0)
            st.upload(x)
            st.apply_ops(circ)
            plain = st.download()
        assert np.array_equal(got, plain)
    want = O.apply_ops_in_place(n, circ, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-4


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
