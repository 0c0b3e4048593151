"""The one-sweep bit permutation (k_permute_bits) without a GPU: the host builds the descriptor
(qip_hip_debug_permute_plan), a numpy restatement of the kernel's index arithmetic replays it.

Checked: every element lands where new[j] = old[src(j)] says; the LDS slots of a tile are a bijection; the lanes of
one LDS bank group (8 lanes for 16-byte elements, 16 for 8-byte ones) hit distinct banks on the write side and on
the read side; composing Swap ops into a permutation equals applying them one after the other (the oracle)."""
import json

import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd import _ffi


def plan(n, pi, row_bits, fold_bits):
    import ctypes as C

    arr = (C.c_uint32 * n)(*pi)
    txt = _ffi.lib.qip_hip_debug_permute_plan(n, arr, row_bits, fold_bits)
    assert txt, _ffi.last_error()
    return json.loads(txt.decode())


def spread(v, positions, first=0):
    """bit (first + i) of v -> position positions[first + i]"""
    out = np.zeros_like(v, dtype=np.uint64)
    for i in range(first, len(positions)):
        out |= ((v >> np.uint64(i)) & np.uint64(1)) << np.uint64(positions[i])
    return out


def fold(c, d):
    f = np.zeros_like(c)
    for a, b in zip(d["fold_from"], d["fold_to"]):
        f ^= ((c >> np.uint64(a)) & np.uint64(1)) << np.uint64(b)
    return c ^ f


def replay(n, d, x):
    """numpy model of k_permute_bits<A, R, TB>: returns (out, slots of the load side [u], slots of the store side [c])"""
    R = d["row_bits"]
    TB = d["tile_bits"]
    nblk = 1 << (n - TB)
    dbase = np.arange(nblk, dtype=np.uint64)
    for p in d["tsorted"]:
        low = dbase & np.uint64((1 << p) - 1)
        dbase = ((dbase >> np.uint64(p)) << np.uint64(p + 1)) | low
    sbase = np.zeros_like(dbase)
    for a, b in zip(d["outer_dst"], d["outer_src"]):
        sbase |= ((dbase >> np.uint64(a)) & np.uint64(1)) << np.uint64(b)
    u = np.arange(1 << TB, dtype=np.uint64)  # e * 256 + t on the load side, the same range on the store side
    row = u & np.uint64((1 << R) - 1)
    s_off = row | spread(u, d["sbits"], R)
    c_of_u = spread(u, d["u2c"])
    slot_ld = fold(c_of_u, d)
    slot_st = fold(u, d)
    d_off = row | spread(u, d["tbits"], R)
    out = np.empty_like(x)
    for b in range(nblk):
        lds = np.empty(1 << TB, dtype=x.dtype)
        lds[slot_ld.astype(np.int64)] = x[(sbase[b] | s_off).astype(np.int64)]
        out[(dbase[b] | d_off).astype(np.int64)] = lds[slot_st.astype(np.int64)]
    return out, slot_ld, slot_st


def want_of(n, pi, x):
    j = np.arange(1 << n, dtype=np.uint64)
    src = np.zeros_like(j)
    for dbit in range(n):
        src |= ((j >> np.uint64(dbit)) & np.uint64(1)) << np.uint64(pi[dbit])
    return x[src.astype(np.int64)]


@pytest.mark.parametrize("row_bits,fold_bits", [(5, 3), (6, 4), (0, 3)])
def test_descriptor_replayed_with_numpy(row_bits, fold_bits):
    """(0, 3) is the shape the library launches for 16-byte elements: 512-byte rows, thread bit 5 = index position 11 on both sides"""
    rng = np.random.default_rng(7)
    split = row_bits == 0
    TB = 12 if split else 2 * row_bits
    row_bits = row_bits or 5
    cases = []
    for n in (TB, TB + 1, TB + 3):
        cases.append((n, list(range(n))[::-1]))                      # bit reversal (QFT's closing swaps)
        cases.append((n, [1, 0] + list(range(2, n))))                # inside a row
        cases.append((n, list(range(1, n)) + [0]))                   # rotation
        cases.append((n, [n - 1] + list(range(1, n - 1)) + [0]))     # one transposition, lowest <-> highest
        for _ in range(6):
            cases.append((n, [int(v) for v in rng.permutation(n)]))
        # only high bits move: rows stay rows on both sides
        hi = list(range(row_bits, n))
        cases.append((n, list(range(row_bits)) + [int(v) for v in rng.permutation(hi)]))
    group = 8 if fold_bits == 3 else 16
    for n, pi in cases:
        d = plan(n, pi, 0 if split else row_bits, fold_bits)
        assert d["tbits"][:row_bits] == list(range(row_bits)) and d["sbits"][:row_bits] == list(range(row_bits))
        if split:
            assert d["split"] == 11 and d["tbits"][5] == 11 and d["sbits"][5] == 11 and d["tile_bits"] in (11, 12)
            assert d["tile_bits"] == 11 or sum(1 for b in range(n) if b < 5 or b == 11 or pi[b] < 5 or pi[b] == 11) == 12
        else:
            assert d["split"] == 0 and d["tile_bits"] == TB
        TB = d["tile_bits"]
        x = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
        got, slot_ld, slot_st = replay(n, d, x)
        assert np.array_equal(got, want_of(n, pi, x)), (n, pi)
        assert len(set(slot_ld.tolist())) == 1 << TB and len(set(slot_st.tolist())) == 1 << TB
        # banks: consecutive lanes of a bank group -> distinct slots modulo the group size
        for slots in (slot_ld, slot_st):
            g = (slots % np.uint64(group)).reshape(-1, group)
            assert all(len(set(r.tolist())) == group for r in g), (n, pi)


def test_run_of_swaps_composes_to_one_permutation():
    """pi after a run of Swap ops: pi_new[d] = pi_old[tau[d]] (the rule schedule_tiles uses), against the oracle"""
    from oracle import qip_oracle as O

    n = 11
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    ops = [q.make_swap_op([0], [10]), q.make_swap_op([1, 2], [9, 4]), q.make_swap_op([10], [3]), q.make_swap_op([5, 6, 7], [8, 0, 1])]
    pi = list(range(n))
    for op in ops:
        h = len(op.indices) // 2
        tau = list(range(n))
        for a, b in zip(op.indices[:h], op.indices[h:]):
            pa, pb = n - 1 - a, n - 1 - b
            tau[pa], tau[pb] = pb, pa
        pi = [pi[tau[dbit]] for dbit in range(n)]
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.array_equal(want_of(n, pi, x), want)


@pytest.mark.parametrize("n,row_bits,fold_bits", [(30, 5, 3), (33, 5, 3), (34, 6, 4), (40, 5, 3), (30, 0, 3), (33, 0, 3), (40, 0, 3)])
def test_descriptor_at_bench_sizes_on_sampled_elements(n, row_bits, fold_bits):
    """no state: for sampled (block, element) pairs the source index the load side computes and the destination index the
    store side computes satisfy dst bit d = src bit pi[d] (block ids above 2^32 blocks included: the 2-D grid of n >= 38)"""
    rng = np.random.default_rng(n)
    for pi in (list(range(n))[::-1], [int(v) for v in rng.permutation(n)], list(range(1, n)) + [0]):
        d = plan(n, pi, row_bits, fold_bits)
        TB, row_bits = d["tile_bits"], d["row_bits"]
        blocks = np.concatenate([np.array([0, (1 << (n - TB)) - 1], dtype=np.uint64), rng.integers(0, 1 << (n - TB), size=200, dtype=np.uint64)])
        dbase = blocks.copy()
        for p in d["tsorted"]:
            low = dbase & np.uint64((1 << p) - 1)
            dbase = ((dbase >> np.uint64(p)) << np.uint64(p + 1)) | low
        sbase = np.zeros_like(dbase)
        for a, b in zip(d["outer_dst"], d["outer_src"]):
            sbase |= ((dbase >> np.uint64(a)) & np.uint64(1)) << np.uint64(b)
        u = rng.integers(0, 1 << TB, size=blocks.size, dtype=np.uint64)  # a source-side coordinate per sampled block
        src = sbase | (u & np.uint64((1 << row_bits) - 1)) | spread(u, d["sbits"], row_bits)
        c = spread(u, d["u2c"])  # where that element sits in the tile = the store side's coordinate
        dst = dbase | (c & np.uint64((1 << row_bits) - 1)) | spread(c, d["tbits"], row_bits)
        for dbit in range(n):
            assert np.array_equal((dst >> np.uint64(dbit)) & np.uint64(1), (src >> np.uint64(pi[dbit])) & np.uint64(1)), (n, dbit)
        assert int(dst.max()) < (1 << n) and int(src.max()) < (1 << n)


def test_plan_hook_rejects_shapes_its_tables_cannot_hold():
    """the exported hook indexes fixed-size tables with its two shape arguments: out-of-range values are errors, not writes
    past the descriptor"""
    import ctypes as C

    n = 16
    arr = (C.c_uint32 * n)(*range(n))
    arr[0], arr[9] = 9, 0
    for row_bits, fold_bits in ((7, 3), (5, 5), (1, 3), (40, 2), (0, 5)):
        assert not _ffi.lib.qip_hip_debug_permute_plan(n, arr, row_bits, fold_bits), (row_bits, fold_bits)
        assert "must be" in _ffi.last_error() or "does not fit" in _ffi.last_error()
    assert _ffi.lib.qip_hip_debug_permute_plan(n, arr, 5, 3)
    assert _ffi.lib.qip_hip_debug_permute_plan(n, arr, 0, 3)  # row_bits = 0: the library's own choice (split rows)


def replay_pairs(n, d, x):
    """numpy model of k_permute_pairs<TB>: 8-byte elements, 16-byte pairs on both global sides, the 8-byte transposition in LDS.
    Returns (out, LDS slots written by the load side [unit, half], LDS slot pairs read by the store side)."""
    TB = d["tile_bits"]
    nblk = 1 << (n - TB)
    dbase = np.arange(nblk, dtype=np.uint64)
    for p in d["tsorted"]:
        low = dbase & np.uint64((1 << p) - 1)
        dbase = ((dbase >> np.uint64(p)) << np.uint64(p + 1)) | low
    sbase = np.zeros_like(dbase)
    for a, b in zip(d["outer_dst"], d["outer_src"]):
        sbase |= ((dbase >> np.uint64(a)) & np.uint64(1)) << np.uint64(b)
    u = np.arange(1 << (TB - 1), dtype=np.uint64) << np.uint64(1)  # a unit's coordinate: (e, thread) above the pair bit
    s_off = spread(u, d["sbits"])          # element offset of the unit's first element on the source side
    d_off = spread(u, d["tbits"])
    assert not np.any(s_off & np.uint64(1)) and not np.any(d_off & np.uint64(1))  # units are 16-byte aligned on both sides
    c_lo = spread(u, d["u2c"])
    slot_lo = fold(c_lo, d)
    slot_hi = slot_lo ^ fold(np.uint64(1) << np.uint64(d["u2c"][0]), d)
    slot_st = fold(u, d)
    assert not np.any(slot_st & np.uint64(1))  # a stored pair is one aligned 16-byte LDS read
    assert fold(np.uint64(1), d) == np.uint64(1)
    out = np.empty_like(x)
    for b in range(nblk):
        lds = np.empty(1 << TB, dtype=x.dtype)
        src = (sbase[b] | s_off).astype(np.int64)
        lds[slot_lo.astype(np.int64)] = x[src]
        lds[slot_hi.astype(np.int64)] = x[src + 1]
        dst = (dbase[b] | d_off).astype(np.int64)
        out[dst] = lds[slot_st.astype(np.int64)]
        out[dst + 1] = lds[slot_st.astype(np.int64) + 1]
    return out, np.stack([slot_lo, slot_hi]), slot_st


def test_pair_form_for_8_byte_elements_replayed_with_numpy():
    """r6, k_permute_pairs (Complex<f32> when index bit 0 moves): the host's descriptor replayed — every element lands where
    new[j] = old[src(j)] says, the LDS slots are a bijection, a wave's 32-lane halves spread their 8-byte writes over >= 16 of the
    32 bank pairs (the sweep is HBM-bound by a factor of ten: two-way conflicts cost nothing, 32-way ones would), and the
    permutations that need 14 tile bits are left to k_permute_bits."""
    rng = np.random.default_rng(11)
    fitted = 0
    for n in (14, 15, 17):
        cases = [[n - 1] + list(range(1, n - 1)) + [0],                     # bit 0 <-> the top bit
                 list(range(1, n)) + [0],                                   # rotation
                 [1, 0] + list(range(2, n)),                                # inside the pair / lane bits
                 [12] + list(range(1, 12)) + [0] + list(range(13, n)),      # bit 0 <-> the split position
                 list(range(n))[::-1]]
        cases += [[int(v) for v in rng.permutation(n)] for _ in range(10)]
        for pi in cases:
            d = plan(n, pi, 100, 0)
            special = [b for b in range(n) if b <= 5 or b == 12 or pi[b] <= 5 or pi[b] == 12]
            assert d["fits"] == (len(special) <= 13), (n, pi)
            if not d["fits"]:
                continue
            fitted += 1
            TB = d["tile_bits"]
            assert TB == (12 if len(special) <= 12 else 13)
            assert d["tbits"][:7] == [0, 1, 2, 3, 4, 5, 12] and d["sbits"][:7] == [0, 1, 2, 3, 4, 5, 12]
            assert all(1 <= t <= 5 for t in d["fold_to"]) and all(f >= 6 for f in d["fold_from"])
            x = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
            got, slots_ld, slot_st = replay_pairs(n, d, x)
            assert np.array_equal(got, want_of(n, pi, x)), (n, pi)
            assert len(set(slots_ld.ravel().tolist())) == 1 << TB
            assert len(set(slot_st.tolist()) | set((slot_st + np.uint64(1)).tolist())) == 1 << TB
            for half in slots_ld:  # 32 consecutive lanes of one write instruction
                g = (half % np.uint64(32)).reshape(-1, 32)
                assert min(len(set(r.tolist())) for r in g) >= 16, (n, pi)
    assert fitted >= 20
    assert plan(20, list(range(20))[::-1], 100, 0)["fits"] is False  # the reversal needs 14 positions
    n = 13
    assert plan(n, list(range(1, n)) + [0], 100, 0)["fits"] is False  # below 14 qubits: k_permute_bits
