"""fusion, tile sweeps (interpreter, run-time-compiled, wide, relabelled), hipGraph programs.
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


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
        if tuning():
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
        c = _ffi.jit_counters()
        return int(c["kernels_resident_total"]), c["compile_ms"] + c["disk_load_ms"]

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
    repeatedly; circuits with an out-of-place op are graphs too (r6: one recording per buffer parity, the host follows the buffers)."""
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
        # a sparse op on 6 qubits with FIVE entries in a row takes the out-of-place literal kernel, with two entries per row it is
        # applied in place through k_sparse_tile where the state is large enough (r4), else out of place through k_sparse_ell: r6 —
        # the program is a graph in every case (an odd number of out-of-place launches: the second run starts on the other buffer
        # and gets a recording of its own; the third run replays the first)
        pos = [n - 1 - qb for qb in perm[:6]]
        kh = sum(1 for pp in pos if not (pp < 5 or pp == (11 if n >= 12 else 5)))  # the op's positions outside the wave row
        tile_form = 3 <= kh <= 7 and n >= 6 + kh + 2
        for width in (5, 2):
            rows = [[((r * 5 + 1 + 7 * e) % 64, 0.5j if e == 0 else 0.25 * (e + 1)) for e in range(width - 1)] + [(r, 2.0)] for r in range(64)]
            sp = circ[:20] + [q.make_sparse_matrix_op(perm[:6], rows)] + circ[20:40]
            with q.HipState(n) as st:
                st.upload(x)
                prog = st.compile_program(sp)
                for _ in range(4):
                    prog.run()
                    assert prog.is_graph, (n, width, kh, tile_form)
                got = st.download()
            want = O.apply_ops_in_place(n, sp + sp + sp + sp, x.copy())
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
        ref.set_option("tile_auto", 0)  # the interpreter, whatever the caches hold (r6: with tile_auto on, apply_ops would take compiled
        ref.upload(x)                   # sweeps when a disk cache already has the whole plan: test_one_shot_apply_ops_... covers that)
        c0 = _ffi.jit_counters()
        ref.apply_ops(ops)
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
        # the state's own apply_ops compiles nothing either (r6: it may REUSE what is resident — the program's segments are its plan's)
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
        # a program whose schedule holds a permutation sweep (out of place) is a graph as well (r6) and stays right
        from rustqip_amd.state import HipProgram

        prog = HipProgram(st, ops)
        for _ in range(2):
            st.upload(x)
            prog.run()
            assert prog.is_graph and np.array_equal(st.download(), plain)
        prog.close()
    xf = rand_state(n, 6, np.complex64)
    with q.HipState(n, np.complex64) as st:
        st.set_option("tile", 1)
        st.upload(xf)
        st.apply_ops(rev)
        assert np.array_equal(st.download(), O.apply_ops_in_place(n, rev, xf.copy()))


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
                if tuning():
                    ref.set_option("tile_jit", 3)  # from here on the reference is the run-time-compiled form with literal numbers
        if tuning():
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
                st.set_option("tile_jit", 1 + 2 * (tid % 2) if tuning() else 1)  # (3 = numbers as literals: a tuning build's value)
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
        if name == "c4" and f64 and tuning():  # global option tile_wide_pin (register pins after block-uniform branches: no semantics; default on): the very same bits without
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
        if name == "grover_k3" and tuning():
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


@pytest.mark.parametrize("dtype,tol", [(np.complex128, TOL64), (np.complex64, TOL32)])
def test_programs_with_this_rounds_kernels(O, dtype, tol):
    """r6: the kernels added this round inside a PROGRAM (tables packed once into the program's own device pool, launches recorded
    into a hipGraph) on a state of 23 qubits, where the pipelined k = 5 tile kernel applies: dense k = 5 complex (four products) and
    real (two), dense k = 6 complex (three products) and real (two), a Swap run that composes to a bit permutation with index bit 0
    moving (k_permute_bits for Complex<f64>, k_permute_pairs for Complex<f32>), a SparseMatrix on seven qubits with four entries per
    row (Complex<f32>: the table in LDS) — replayed twice, against the same ops applied one call at a time and against the oracle."""
    n = 23
    rng = np.random.default_rng(66)
    P = lambda *pos: [n - 1 - p for p in pos]  # noqa: E731
    rows7 = []
    for r in range(128):
        cols = [int(c) for c in rng.permutation(128)[:4]]
        rows7.append([(c, complex(rng.standard_normal(), rng.standard_normal()) * 0.5) for c in cols])
    ops = [
        q.make_matrix_op(P(22, 3, 11, 7, 5), rand_unitary(5, rng).ravel()),
        q.make_matrix_op(P(20, 0, 12, 6, 14), np.linalg.qr(rng.standard_normal((32, 32)))[0].ravel()),
        q.make_matrix_op(P(21, 1, 5, 11, 8, 15), rand_unitary(6, rng).ravel()),
        q.make_matrix_op(P(2, 9, 13, 17, 19, 22), np.linalg.qr(rng.standard_normal((64, 64)))[0].ravel()),
        q.make_swap_op(P(0, 13), P(22, 4)),
        q.make_sparse_matrix_op(P(22, 18, 16, 14, 9, 3, 1), rows7),
        q.make_control_op(P(15), q.make_matrix_op(P(22, 0, 5, 11, 13), rand_unitary(5, rng).ravel())),
    ]
    x = circuits.random_state(n, seed=9, dtype=dtype)
    with q.HipState(n, dtype) as st, q.HipState(n, dtype) as eager:
        st.upload(x)
        prog = st.compile_program(ops)  # (ops carry complex128 tables; the binding converts them per state dtype)
        prog.run()
        prog.run()
        assert prog.is_graph
        got = st.download()
        prog.close()
        eager.upload(x)
        eager.set_option("profile", 1)
        for _ in range(2):
            for o in ops:
                eager.apply_op(o)
        prof = eager.profile()
        assert np.array_equal(got, eager.download())  # the same kernels with the same tables: identical
        assert prof.get("k_gate_kq_mfma", {}).get("launches", 0) >= 6 and prof.get("k_gate_big_mfma", {}).get("launches", 0) == 4, prof
    want = O.apply_ops_in_place(n, ops + ops, x.copy())
    scale = max(1.0, float(np.max(np.abs(want))))  # (the sparse op is not unitary: amplitudes grow)
    assert float(np.max(np.abs(got - want))) <= 8 * tol * scale


def test_programs_own_their_payloads_and_record_out_of_place_ops(O):
    """r6: a program packs every op's tables ONCE into device memory it owns and records kernel nodes only;
    ops that write the second buffer are recorded too — one recording per starting buffer, the host follows the ping-pong
    (builder.rs:514).  The reference's own bench shapes that used to fall back to eager (re-pack + re-upload per application):
    state_bench.rs:118-139 (n = 8, dense 8-qubit gate: the literal kernel at that size) and :380-393 (n = 16, 16-qubit sparse
    identity), plus a 16-qubit sparse PERMUTATION with phases so that a wrong buffer would show."""
    rng = np.random.default_rng(5)
    s2 = math.sqrt(0.5)
    h8 = np.array([[1.0]])
    for _ in range(8):
        h8 = np.kron(h8, np.array([[s2, s2], [s2, -s2]]))
    perm16 = rng.permutation(1 << 16)
    phase16 = np.exp(1j * rng.uniform(0, 6, 1 << 16))
    shapes = [
        ("dense8_n8", 8, q.make_matrix_op(list(range(8)), h8.ravel())),
        ("dense8_unitary_n9", 9, q.make_matrix_op([8, 0, 3, 1, 5, 2, 7, 4], rand_unitary(8, rng).ravel())),
        ("sparse16_identity", 16, q.make_sparse_matrix_op(list(range(16)), [[(i, 1.0)] for i in range(1 << 16)])),
        ("sparse16_permutation", 16, q.make_sparse_matrix_op(list(range(16)), [[(int(perm16[i]), complex(phase16[i]))] for i in range(1 << 16)])),
        ("sparse15_two_per_row_n17", 17, q.make_sparse_matrix_op(list(range(1, 16)), [[(i, 0.6), (i ^ 0x1234, 0.8j)] for i in range(1 << 15)])),
    ]
    for name, n, op in shapes:
        x = rand_state(n, 11)
        for reps in (1, 2, 3):  # odd counts end on the other buffer: the next run needs the second recording
            with q.HipState(n) as st:
                st.upload(x)
                prog = st.compile_program([op] * reps)
                runs = 0
                for _ in range(3):
                    prog.run()
                    runs += reps
                    assert prog.is_graph, (name, reps)
                # an eager out-of-place op between two runs exchanges the buffers under the program: it picks the other recording
                st.apply_op(op)
                prog.run()
                assert prog.is_graph, (name, reps)
                runs += 1 + reps
                got = st.download()
                prog.close()
            want = O.apply_ops_in_place(n, [op] * runs, x.copy())
            if name.startswith("dense"):  # (k_dense_small: partial sums of 16 columns, the 1e-12 bar of dense k >= 3 gates)
                assert float(np.max(np.abs(got - want))) <= TOL64 * max(1.0, float(np.max(np.abs(want)))), (name, reps)
            else:
                assert np.array_equal(got, want), (name, reps, float(np.max(np.abs(got - want))))
    # the payload lives on the device: the host table may change after the program was created (a graph replays nothing from the host)
    n = 12
    x = rand_state(n, 12)
    u = rand_unitary(2, rng)
    d3 = np.diag(np.exp(1j * rng.uniform(0, 6, 8)))
    ops = [q.make_matrix_op([3, 9], u.ravel()), q.make_matrix_op([0, 5, 11], d3.ravel())]
    want = O.apply_ops_in_place(n, ops + ops, x.copy())
    with q.HipState(n) as st:
        st.upload(x)
        prog = st.compile_program(ops)
        prog.run()
        for cop in prog._compiled[1]:
            for buf in cop._keep:
                if isinstance(buf, np.ndarray) and buf.dtype == np.complex128:
                    buf[:] = 0
        prog.run()
        assert prog.is_graph and np.array_equal(st.download(), want)
        prog.close()


@pytest.mark.slow
def test_one_shot_apply_ops_takes_compiled_sweeps_only_when_they_are_free(O, tmp_path):
    """r6, option tile_auto for `calculate_state`-style callers: apply_ops on a state with tile = 1 and
    tile_jit = 0 runs the interpreter when a segment of its (wide) plan is neither resident nor in the disk cache — and hands the
    misses to background helpers; once they are there (a later call, another process) the same call runs the compiled sweeps.
    Bit-identical either way."""
    import time

    from rustqip_amd import _ffi

    n = 24  # (the smallest state whose one-shot apply_ops looks its plan up: below, the lookup costs more than the sweeps save)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 96, seed=41)
    x = circuits.random_state(n, seed=4)
    assert _ffi.lib.qip_hip_jit_set_cache_dir(str(tmp_path / "cache").encode()) == 0
    try:
        with q.HipState(n) as ref:  # the interpreter, whatever the cache holds
            ref.set_option("tile", 1)
            ref.set_option("tile_auto", 0)
            ref.upload(x)
            ref.apply_ops(ops)
            want = ref.download()
        assert np.array_equal(want, O.apply_ops_in_place(n, ops, x.copy()))
        c0 = _ffi.jit_counters()
        with q.HipState(n) as st:
            st.set_option("tile", 1)
            st.upload(x)
            t0 = time.perf_counter()
            st.apply_ops(ops)  # cold: the interpreter now, the plan's segments to the background
            st.sync()
            cold_s = time.perf_counter() - t0
            c1 = _ffi.jit_counters()
            assert np.array_equal(st.download(), want)
            assert c1["kernels_resident_total"] == c0["kernels_resident_total"] and c1["compiled"] == c0["compiled"]
            segs = c1["background_segments"] - c0["background_segments"]
            assert segs >= 2 and cold_s < 5.0, (segs, cold_s)  # nobody waited for a compiler
            st.upload(x)
            st.apply_ops(ops)  # asked again at once: still the interpreter, nothing handed out twice
            assert _ffi.jit_counters()["background_segments"] == c1["background_segments"]
            assert np.array_equal(st.download(), want)
            deadline = time.time() + 180
            while time.time() < deadline and len([f for f in os.listdir(tmp_path / "cache") if f.endswith(".co")]) < segs:
                time.sleep(0.5)
            assert len([f for f in os.listdir(tmp_path / "cache") if f.endswith(".co")]) == segs
            time.sleep(1.6)  # (a plan seen incomplete is not looked up again for 1.5 s)
            st.upload(x)
            st.set_option("profile", 0)
            st.apply_ops(ops)  # warm: every segment is a disk hit -> compiled wide sweeps
            c2 = _ffi.jit_counters()
            assert c2["disk_hits"] - c1["disk_hits"] == segs and c2["kernels_resident_total"] - c1["kernels_resident_total"] == segs
            assert c2["compiled"] == c1["compiled"]  # (this process compiled nothing)
            assert np.array_equal(st.download(), want)
    finally:
        assert _ffi.lib.qip_hip_jit_set_cache_dir(None) == 0
