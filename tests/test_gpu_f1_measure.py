"""measurement on the device.
Split out of the former tests/test_parity_gpu.py (a `-x` failure now names the row).  Everything goes through the
C ABI (ctypes -> libqip_hip.so -> HIP kernels); helpers and bars: tests/gpu_common.py."""
from gpu_common import *  # noqa: F401,F403
from gpu_common import _ansatz, _jit_info, _permuted, _run_dist, _special_gates  # noqa: F401

pytestmark = pytest.mark.gpu


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
            one = two
            if tuning():  # (the one-launch variant: measured slower, a tuning build's option)
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
