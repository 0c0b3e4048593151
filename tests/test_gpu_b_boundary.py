"""SURVEY.md §8 row b (+ f2): the drop-in boundary — host twins, error behaviour, two states side by side, circuit replay.
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


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
