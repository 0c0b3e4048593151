"""The host half of the tile sweeps, without a GPU: qip_hip_debug_tile_plan gives the schedule, the passes and the
gate descriptors exactly as they are shipped to k_tile_passes; a numpy model of that kernel (this file: the
kernel's semantics restated, tile by tile, pass by pass, lane by lane) replays them on a CPU state vector and the
result must match the oracle applying the circuit gate by gate.  Covers: scheduling legality (commutation rules),
free-bit selection, pass grouping, the lane-bit assignment, the dispatch codes and pass-bit indices the host
resolves, the split of controls into pass-bit / lane-bit / outside-the-tile parts, 2-qubit matrix order."""
import cmath
import math

import numpy as np
import pytest

import rustqip_amd as q
from oracle import qip_oracle as O
from rustqip_amd import _ffi, circuits
from rustqip_amd.ops import TILE_BITS, TILE_LANE_BITS, debug_tile_plan

TILE_LOW, OUTSIDE = 6, 0xFFFFFFFF
NLANES = 1 << TILE_LANE_BITS
# enum TileOp (rustqip_amd/csrc/qip_kernels.h)
DIAG_UNIFORM, DIAG_LANE, DIAG_LANE_CTL, DIAG_REG0 = 0, 1, 2, 3
DENSE0, DENSE_LANE0, DENSE2Q_FIRST, SWAP_FIRST = 6, 9, 12, 18
DENSE3Q_FIRST = 21
DENSE2Q = [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)]
SWAPS = [(0, 1), (0, 2), (1, 2)]


def emulate_segment(state, n, seg, use_interp=False):
    """k_tile_passes on a numpy vector: one block per tile, 512 lanes, 8 elements per lane and pass."""
    high = seg["high"]
    # the six low tile bits = the lane id at load / store time: positions 0..4 and p5 (5: one contiguous 1-KiB row per wave-level
    # access; 11: two 512-byte halves 32 KiB apart, the r4 default for Complex<f64>), exported by the plan as "low"
    low = seg.get("low", list(range(TILE_LOW)))
    assert len(low) == TILE_LOW and low[:5] == [0, 1, 2, 3, 4] and low[5] in (5, 11) and not set(low) & set(high)
    tile_pos = low + high
    other = [p for p in range(n) if p not in tile_pos]  # the block index fills these, ascending (insert_bits)
    ntiles = 1 << (n - TILE_BITS)
    bid = np.arange(ntiles, dtype=np.uint64)
    base = np.zeros(ntiles, dtype=np.uint64)
    for k, p in enumerate(other):
        base |= ((bid >> np.uint64(k)) & np.uint64(1)) << np.uint64(p)
    t = np.arange(1 << TILE_BITS, dtype=np.uint64)
    off = np.zeros_like(t)
    for b, p in enumerate(tile_pos):
        off |= ((t >> np.uint64(b)) & np.uint64(1)) << np.uint64(p)
    idx = (base[:, None] | off[None, :]).astype(np.int64)
    tile = state[idx]  # (ntiles, 2^TILE_BITS): the LDS-resident tile, indexed by tile index
    gates = seg["gates"]
    mats = np.array([complex(a, b) for a, b in seg["mats"]], dtype=np.complex128)
    tid = np.arange(NLANES, dtype=np.int64)
    seen = 0
    interp = seg.get("interp") if use_interp else None
    pass_no = 0
    for ps in seg["passes"]:
        pb, lanepos = ps["pb"], ps["lanepos"]
        assert sorted(pb + lanepos) == list(range(TILE_BITS))  # lane bits + pass bits tile the tile bits exactly
        tb = np.zeros(NLANES, dtype=np.int64)
        for k in range(TILE_LANE_BITS):
            tb |= ((tid >> k) & 1) << lanepos[k]
        c = np.array([sum(((i >> j) & 1) << pb[j] for j in range(3)) for i in range(8)], dtype=np.int64)
        te = tb[:, None] | c[None, :]  # (lanes, 8): a bijection onto the tile indices
        assert np.array_equal(np.sort(te.ravel()), np.arange(1 << TILE_BITS))
        e = tile[:, te]  # (ntiles, lanes, 8)
        assert ps["first"] == seen
        seen += ps["count"]
        pass_gates = gates[ps["first"]: ps["first"] + ps["count"]]
        if interp is not None:  # r5: what the interpreter kernel is handed — runs of diagonal gates as TileDiagItem steps
            first, count = interp["passes"][pass_no]
            pass_gates = interp["gates"][first: first + count]
        pass_no += 1
        for g in pass_gates:
            if "run" in g:  # TOP_DIAG_RUN (qip_kernels.h): F = (tb & sel) ? f1 : f0; (1, 0) where a lane condition fails; elements by reg
                for it in interp["items"][g["run"][0]: g["run"][0] + g["run"][1]]:
                    tile_on = (base & np.uint64(it["out"][0])) == np.uint64(it["out"][1])
                    f0, f1 = complex(*it["f0"]), complex(*it["f1"])
                    f = np.where((tb & it["sel"]) != 0, f1, f0) if it["sel"] else np.full(NLANES, f1)
                    if it["lane"][0]:
                        f = np.where((tb & it["lane"][0]) == it["lane"][1], f, 1.0 + 0j)
                    elem_ok = (c & it["reg"][0]) == it["reg"][1]
                    assert sum(1 << i for i in range(8) if elem_ok[i]) == it["emask"]  # what the kernel tests: the host-resolved element mask
                    mask = tile_on[:, None, None] & elem_ok[None, None, :]
                    e = np.where(mask, f.astype(e.dtype)[None, :, None] * e, e)  # (the state's precision, like the scalar factors of the per-op paths)
                continue
            if "gate" in g:
                g = gates[g["gate"]]
            passmask = sum(1 << b for b in pb)
            assert g["cm_reg"] == g["cmask"] & passmask and g["cm_lane"] == g["cmask"] & ~passmask
            m = [complex(a, b) for a, b in g["m"]]
            tile_on = (base & np.uint64(g["omask"])) == np.uint64(g["omask"])  # outside controls: per block
            elem_ok = (c & g["cm_reg"]) == g["cm_reg"]
            lane_ok = (tb & g["cm_lane"]) == g["cm_lane"]
            mask = tile_on[:, None, None] & lane_ok[None, :, None] & elem_ok[None, None, :]
            op = g["op"]
            if op in (DIAG_UNIFORM, DIAG_LANE, DIAG_LANE_CTL):
                assert g["kind"] == 1 and (op != DIAG_UNIFORM or (g["b0"] == OUTSIDE and g["cm_lane"] == 0))
                assert (op == DIAG_LANE_CTL) == (g["cm_lane"] != 0) or op == DIAG_UNIFORM
                if g["b0"] == OUTSIDE:
                    one = np.broadcast_to((((base >> np.uint64(g["tpos_out"])) & np.uint64(1)) != 0)[:, None], (ntiles, NLANES))
                else:
                    assert not (passmask >> g["b0"]) & 1
                    one = np.broadcast_to((((tb >> g["b0"]) & 1) != 0)[None, :], (ntiles, NLANES))
                f = np.where(one, m[1], m[0])
                e = np.where(mask, f[:, :, None] * e, e)
            elif DIAG_REG0 <= op < DIAG_REG0 + 3:
                j = op - DIAG_REG0
                assert g["kind"] == 1 and g["b0"] == pb[j]
                for i in range(8):
                    e[:, :, i] = np.where(mask[:, :, i], m[(i >> j) & 1] * e[:, :, i], e[:, :, i])
            elif DENSE0 <= op < DENSE0 + 6:
                j = (op - DENSE0) % 3
                assert g["kind"] == 0 and g["b0"] == pb[j] and ((op >= DENSE_LANE0) == (g["cm_lane"] != 0))
                for i in range(8):
                    if (i >> j) & 1:
                        continue
                    k = i | (1 << j)
                    a0, a1 = e[:, :, i].copy(), e[:, :, k].copy()
                    e[:, :, i] = np.where(mask[:, :, i], m[0] * a0 + m[1] * a1, a0)
                    e[:, :, k] = np.where(mask[:, :, i], m[2] * a0 + m[3] * a1, a1)
            elif DENSE2Q_FIRST <= op < DENSE2Q_FIRST + 6:
                ja, jb = DENSE2Q[op - DENSE2Q_FIRST]
                assert g["kind"] == 3 and g["b0"] == pb[ja] and g["b1"] == pb[jb]
                mat = mats[16 * g["nz"]: 16 * g["nz"] + 16].reshape(4, 4)
                jc = 3 - ja - jb
                for qd in range(2):
                    ids = [(qd << jc) | ((s >> 1) << ja) | ((s & 1) << jb) for s in range(4)]
                    x = np.stack([e[:, :, i] for i in ids], axis=-1)
                    y = x @ mat.T
                    for r, i in enumerate(ids):
                        e[:, :, i] = np.where(mask[:, :, ids[0]], y[:, :, r], x[:, :, r])
            elif DENSE3Q_FIRST <= op < DENSE3Q_FIRST + 6:
                ja, jb = DENSE2Q[op - DENSE3Q_FIRST]
                jc = 3 - ja - jb
                assert g["kind"] == 4 and g["b0"] == pb[ja] and g["b1"] == pb[jb] and g["tpos_out"] == pb[jc] and g["cm_reg"] == 0
                mat = mats[16 * g["nz"]: 16 * g["nz"] + 64].reshape(8, 8)
                ids = [(((s_ >> 2) & 1) << ja) | (((s_ >> 1) & 1) << jb) | ((s_ & 1) << jc) for s_ in range(8)]
                x = np.stack([e[:, :, i] for i in ids], axis=-1)
                y = x @ mat.T
                for r, i in enumerate(ids):
                    e[:, :, i] = np.where(mask[:, :, 0], y[:, :, r], x[:, :, r])
            elif SWAP_FIRST <= op < SWAP_FIRST + 3:
                j0, j1 = SWAPS[op - SWAP_FIRST]
                assert g["kind"] == 2 and g["b0"] == pb[j0] and g["b1"] == pb[j1]
                for i in range(8):
                    if ((i >> j0) & 1, (i >> j1) & 1) != (1, 0):
                        continue
                    k = (i & ~(1 << j0)) | (1 << j1)
                    a, b = e[:, :, i].copy(), e[:, :, k].copy()
                    e[:, :, i] = np.where(mask[:, :, i], b, a)
                    e[:, :, k] = np.where(mask[:, :, i], a, b)
            else:
                raise AssertionError(f"unknown dispatch code {op}")
        tile[:, te] = e
    assert seen == len(gates)
    state[idx] = tile


def emulate_wide_segment(state, n, seg):
    """A wide segment (option tile_wide: a 13-bit tile held in registers, rustqip_amd/csrc/qip_tile.h WidePlan) on a numpy vector.
    Two things are checked: the STRUCTURE of the plan — arrangement 0 is the load arrangement, every gate's exchange bits are
    register bits of its pass, consecutive arrangements share the two quarter bits, the buffer layout of every transposition is
    a bijection, the plan returns to the load arrangement — and its MEANING: the gates, in the plan's order, applied in the
    13-bit tile-index space with their tile-bit / outside-the-tile masks (a transposition moves nothing logically)."""
    WB = 13
    low, high = seg["low"], seg["high"]
    tile_pos = low + high
    other = [p for p in range(n) if p not in tile_pos]
    ntiles = 1 << (n - WB)
    bid = np.arange(ntiles, dtype=np.uint64)
    base = np.zeros(ntiles, dtype=np.uint64)
    for k, p in enumerate(other):
        base |= ((bid >> np.uint64(k)) & np.uint64(1)) << np.uint64(p)
    t = np.arange(1 << WB, dtype=np.int64)
    off = np.zeros(1 << WB, dtype=np.uint64)
    for b, p in enumerate(tile_pos):
        off |= ((t.astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(p)
    idx = (base[:, None] | off[None, :]).astype(np.int64)
    tile = state[idx]
    passes, gates = seg["passes"], seg["gates"]
    mats = np.array([complex(a, b) for a, b in seg["mats"]], dtype=np.complex128)
    load_R, load_L = [8, 9, 10, 11, 12], list(range(8))
    assert passes[0]["R"] == load_R and passes[0]["L"] == load_L and not passes[0]["transposed"]
    if len(passes) > 1:
        assert passes[-1]["R"] == load_R and passes[-1]["L"] == load_L
    seen = 0
    for pi, ps in enumerate(passes):
        assert sorted(ps["R"] + ps["L"]) == list(range(WB))
        if pi:
            prev = passes[pi - 1]
            assert ps["transposed"] and ps["q"][0] != ps["q"][1] and set(ps["q"]) <= set(ps["R"]) & set(prev["R"])
            assert len(set(ps["R"]) - set(prev["R"])) <= 3
            nonq = [b for b in range(WB) if b not in ps["q"]]
            assert sorted(ps["bufpos"][b] for b in nonq) == list(range(11))
            # the writing arrangement's thread bits 0..3 on buffer bits 0..3; the reading one's on pairwise distinct classes
            assert [ps["bufpos"][prev["L"][k]] for k in range(4)] == [0, 1, 2, 3]
            cls = [ps["bufpos"][ps["L"][k]] for k in range(4)]
            assert all(c < 8 for c in cls) and len({c & 3 for c in cls}) == 4, cls
        assert ps["first"] == seen or ps["count"] == 0
        for g in gates[ps["first"]: ps["first"] + ps["count"]]:
            kind = g["kind"]
            ex = [] if kind == 1 else [g["b0"]] if kind == 0 else [g["b0"], g["b1"]] if kind in (2, 3) else [g["b0"], g["b1"], g["tpos_out"]]
            assert set(ex) <= set(ps["R"]), (ex, ps["R"])
            tile_on = (base & np.uint64(g["omask"])) == np.uint64(g["omask"])
            ok = (t & g["cmask"]) == g["cmask"]
            m = [complex(a, b) for a, b in g["m"]]
            if kind == 0:
                b = g["b0"]
                i0 = t[ok & (((t >> b) & 1) == 0)]
                a0, a1 = tile[:, i0].copy(), tile[:, i0 | (1 << b)].copy()
                tile[:, i0] = np.where(tile_on[:, None], m[0] * a0 + m[1] * a1, a0)
                tile[:, i0 | (1 << b)] = np.where(tile_on[:, None], m[2] * a0 + m[3] * a1, a1)
            elif kind == 1:
                if g["b0"] == OUTSIDE:
                    f = np.where(((base >> np.uint64(g["tpos_out"])) & np.uint64(1)) != 0, m[1], m[0])[:, None] * np.ones(1 << WB)[None, :]
                else:
                    f = np.where(((t >> g["b0"]) & 1) != 0, m[1], m[0])[None, :] * np.ones(ntiles)[:, None]
                tile = np.where(tile_on[:, None] & ok[None, :], f * tile, tile)
            elif kind == 2:
                b0, b1 = g["b0"], g["b1"]
                i10 = t[ok & (((t >> b0) & 1) == 1) & (((t >> b1) & 1) == 0)]
                i01 = (i10 & ~(1 << b0)) | (1 << b1)
                a, b = tile[:, i10].copy(), tile[:, i01].copy()
                tile[:, i10] = np.where(tile_on[:, None], b, a)
                tile[:, i01] = np.where(tile_on[:, None], a, b)
            else:
                bits = [g["b0"], g["b1"]] if kind == 3 else [g["b0"], g["b1"], g["tpos_out"]]  # sub-index MSB first
                k = len(bits)
                mat = mats[16 * g["nz"]: 16 * g["nz"] + (1 << (2 * k))].reshape(1 << k, 1 << k)
                sel = ok.copy()
                for b in bits:
                    sel &= ((t >> b) & 1) == 0
                i0 = t[sel]
                ids = [i0 | sum(((c >> (k - 1 - j)) & 1) << bits[j] for j in range(k)) for c in range(1 << k)]
                x = np.stack([tile[:, i] for i in ids], axis=-1)
                y = x @ mat.T
                for r, i in enumerate(ids):
                    tile[:, i] = np.where(tile_on[:, None], y[:, :, r], x[:, :, r])
        seen += ps["count"]
    assert seen == len(gates)
    state[idx] = tile


def replay(n, ops, mode, x, dtype=None):
    plan = debug_tile_plan(n, ops, mode) if dtype is None else debug_tile_plan(n, ops, mode, dtype)
    assert plan["n"] == n
    relabelled = "circuit" in plan
    if relabelled:
        # mode bit 2: the steps index the circuit as the scheduler rewrote it — the caller's ops under the qubit labels in
        # force when they run ("o" = position in the caller's circuit, "i" = qubit indices), in-tile swaps it inserted
        # ("o" = -1); uncontrolled Swap ops of the caller became label exchanges and appear nowhere
        import dataclasses

        n_in, kept = len(ops), [c["o"] for c in plan["circuit"] if c["o"] >= 0]
        assert len(set(kept)) == len(kept) and plan["absorbed"] == n_in - len(kept) and plan["inserted"] == sum(c["o"] < 0 for c in plan["circuit"])
        assert all(ops[o].kind == "Swap" for o in set(range(n_in)) - set(kept))
        ops = [q.make_swap_op(c["i"][:1], c["i"][1:]) if c["o"] < 0 else dataclasses.replace(ops[c["o"]], indices=list(c["i"])) for c in plan["circuit"]]
    st = x.copy()
    done = []
    for step in plan["steps"]:
        if "perm" in step:  # a run of Swap ops as one bit permutation: new[j] = old[src(j)], bit perm[d] of src = bit d of j
            assert (len(step["ops"]) >= 2 or relabelled) and sorted(step["perm"]) == list(range(n))
            j = np.arange(1 << n, dtype=np.uint64)
            src = np.zeros_like(j)
            for dbit, sbit in enumerate(step["perm"]):
                src |= ((j >> np.uint64(dbit)) & np.uint64(1)) << np.uint64(sbit)
            st = st[src.astype(np.int64)]
        elif len(step["ops"]) == 1:
            st = O.apply_ops_in_place(n, [ops[step["ops"][0]]], st)
        elif step.get("wide"):
            assert len(step["high"]) == 7 and len(set(step["high"])) == 7 and not set(step["high"]) & set(step["low"])
            emulate_wide_segment(st, n, step)
        else:
            assert len(step["high"]) == TILE_BITS - TILE_LOW and len(set(step["high"])) == TILE_BITS - TILE_LOW
            assert not set(step["high"]) & set(step["low"]) and step["low"][5] == (11 if n >= 12 and step["low"][5] != 5 else 5)
            emulate_segment(st, n, step, use_interp=bool(mode & 1024))
        done += step["ops"]
    assert sorted(done) == list(range(len(ops)))
    return st, plan


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def fuzz_circuit(n, rng, gates):
    s2 = 0.5 ** 0.5
    g1 = [[0, 1, 1, 0], [0, -1j, 1j, 0], [1, 0, 0, -1], [s2, s2, s2, -s2], [1, 0, 0, 1j], [1, 0, 0, cmath.rect(1, 0.785)],
          [cmath.rect(1, -0.35), 0, 0, cmath.rect(1, 0.35)], [1, 1, 0, 1], [0.3 + 0.1j, -0.7j, 0.2, 0.9 - 0.4j]]
    ops = []
    for _ in range(gates):
        perm = [int(v) for v in rng.permutation(n)]
        shape = int(rng.integers(0, 9))
        nc = int(rng.integers(0, 5))
        if shape <= 3:
            g = q.make_matrix_op([perm[0]], g1[int(rng.integers(0, len(g1)))])
            ops.append(q.make_control_op(perm[1:1 + nc], g) if nc and rng.integers(0, 2) else g)
        elif shape == 4:
            g = q.make_matrix_op([perm[0]], [1, 0, 0, cmath.rect(1, float(rng.uniform(0, 6.28)))])
            ops.append(q.make_control_op(perm[1:2 + nc], g))
        elif shape == 5:
            g = q.make_swap_op([perm[0]], [perm[1]])
            ops.append(q.make_control_op(perm[2:2 + nc], g) if nc else g)
        elif shape == 6:
            g = q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel())
            ops.append(q.make_control_op(perm[2:2 + min(nc, 3)], g) if nc else g)
        elif shape == 7:
            g = q.make_matrix_op(perm[:3], rand_unitary(3, rng).ravel())  # a pass of its own three bits
            ops.append(q.make_control_op(perm[3:3 + min(nc, 3)], g) if nc and rng.integers(0, 2) else g)
        else:
            ops.append(q.make_swap_op(perm[:2], perm[2:4]))  # not tileable
    return ops


@pytest.fixture(params=[1, 2], ids=["sched_default", "sched_search"])
def sched(request):
    """global option "tile_sched": 1 (default) tries the rules for claiming a segment's positions (first come / what a
    position buys) from n = 24 up and keeps the shortest plan, 2 does so at every size — the sizes the numpy model replays"""
    q.set_global_option("tile_sched", request.param)
    yield request.param
    q.set_global_option("tile_sched", 1)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", ["c2", "qft", "c4", "grover", "fuzz12", "fuzz13", "fuzz14", "c2n17", "fuzz18"])
def test_tile_plan_replayed_on_cpu_matches_the_oracle(name, mode, sched):
    n = {"fuzz13": 13, "fuzz14": 14, "c2n17": 17, "fuzz18": 18}.get(name, 12)
    rng = np.random.default_rng(len(name) * 7 + n)
    ops = {
        "c2": lambda: circuits.h_layer(n) + circuits.c2_random_circuit(n, 120, seed=28),
        "c2n17": lambda: circuits.c2_random_circuit(n, 200, seed=17),
        "qft": lambda: circuits.c3_qft(n),
        "c4": lambda: circuits.c4_clifford_t(n, 120, seed=32),
        "grover": lambda: circuits.h_layer(n) + circuits.c5_grover_iteration(n),
    }.get(name, lambda: fuzz_circuit(n, rng, 140))()
    x = circuits.random_state(n, seed=n)
    got, plan = replay(n, ops, mode, x)
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, float(np.max(np.abs(want))))
    assert len(plan["steps"]) < len(ops)  # gates really share sweeps
    if sched == 2:  # the search keeps the shortest of its plans, the first-come plan among them
        q.set_global_option("tile_sched", 0)
        assert len(plan["steps"]) <= len(debug_tile_plan(n, ops, mode)["steps"])


def test_diagonal_runs_of_the_interpreter_are_the_same_products_in_the_same_order():
    """r5: the interpreter kernel is handed every run of >= 2 consecutive diagonal gates of a pass as ONE entry + TileDiagItem
    steps (qip_hip_debug_tile_plan mode bit 1024 exports that form).  Replayed with the numpy model: the state after every
    circuit equals the plain plan's replay EXACTLY (same products, same order — numpy multiplies complex numbers the unfused
    way on both sides) and the oracle to rounding; QFT's segments really are mostly runs."""
    rng = np.random.default_rng(12)
    n = 13
    x = circuits.random_state(n, seed=4)
    import cmath

    extra = []
    for _ in range(40):  # diagonal gates of every shape: phase / Rz-like / units on either entry, 0..3 controls anywhere
        qs = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 5))
        d = {0: [1, 0, 0, cmath.rect(1, 0.3)], 1: [cmath.rect(1, -0.2), 0, 0, cmath.rect(1, 0.2)], 2: [cmath.rect(1, 0.9), 0, 0, 1],
             3: [1, 0, 0, -1], 4: [1j, 0, 0, cmath.rect(1, 1.1)]}[kind]
        nc = int(rng.integers(0, 4))
        op = q.make_matrix_op([qs[0]], d)
        extra.append(q.make_control_op(qs[1:1 + nc], op) if nc else op)
        if rng.integers(0, 3) == 0:
            extra.append(q.make_matrix_op([qs[5]], circuits.H))
    cases = {"qft": circuits.c3_qft(n), "c4": circuits.c4_clifford_t(n, 120, seed=3),
             "c2": circuits.h_layer(n) + circuits.c2_random_circuit(n, 100, seed=9), "diag_shapes": circuits.h_layer(n) + extra,
             "grover": circuits.c5_grover_iteration(n)}
    for name, ops in cases.items():
        for mode in (1, 2):
            plain = replay(n, ops, mode, x)[0]
            runs = replay(n, ops, mode | 1024, x)[0]
            assert np.array_equal(plain, runs), (name, mode)
            assert np.max(np.abs(runs - O.apply_ops_in_place(n, ops, x.copy()))) < 1e-12, (name, mode)
    plan = debug_tile_plan(n, cases["qft"], 1 | 1024)
    segs = [s for s in plan["steps"] if "interp" in s]
    in_runs = sum(s["interp"]["gates_in_runs"] for s in segs)
    assert segs and in_runs >= 0.7 * sum(len(s["gates"]) for s in segs), in_runs
    # Complex<f32> states: the same steps with f32 numbers
    xf = x.astype(np.complex64)
    # (held to f32 rounding, not equality: the numpy model of the per-op LANE path promotes its per-lane factor array to f64)
    assert np.max(np.abs(replay(n, cases["diag_shapes"], 1, xf, _ffi.QIP_C32)[0] - replay(n, cases["diag_shapes"], 1 | 1024, xf, _ffi.QIP_C32)[0])) < 1e-6
    assert np.array_equal(replay(n, cases["qft"], 1, xf, _ffi.QIP_C32)[0], replay(n, cases["qft"], 1 | 1024, xf, _ffi.QIP_C32)[0])


def test_runs_of_swaps_become_one_permutation_sweep():
    """QFT's closing bit reversal (n/2 transpositions over every index bit) cannot share a tile segment (a segment holds
    five free positions): the scheduler composes the run into ONE bit-permutation sweep.  Exact: swaps only move data."""
    n = 16
    x = circuits.random_state(n, seed=3)
    ops = [q.make_swap_op([i], [n - 1 - i]) for i in range(n // 2)] + [q.make_matrix_op([5], circuits.T), q.make_matrix_op([3], circuits.H)]
    got, plan = replay(n, ops, 1, x)
    perm_steps = [s for s in plan["steps"] if "perm" in s]
    assert len(perm_steps) == 1 and len(perm_steps[0]["ops"]) == n // 2 and len(plan["steps"]) == 2
    assert perm_steps[0]["perm"] == list(range(n))[::-1]
    assert np.max(np.abs(got - O.apply_ops_in_place(n, ops, x.copy()))) <= 1e-12  # (numpy's complex products round differently)
    got, plan = replay(n, ops[: n // 2], 1, x)
    assert len(plan["steps"]) == 1 and np.array_equal(got, O.apply_ops_in_place(n, ops[: n // 2], x.copy()))
    # inside QFT the segments before the reversal absorb the swaps that fit them; what is left still goes as one sweep
    n = 20
    ops = circuits.c3_qft(n)
    x = circuits.random_state(n, seed=4)
    got, plan = replay(n, ops, 1, x)
    assert sum("perm" in s for s in plan["steps"]) == 1
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-12
    n = 16
    x = circuits.random_state(n, seed=3)
    # a run that fits one segment stays a segment; controlled swaps and runs broken by another gate are left alone
    rng = np.random.default_rng(5)
    for trial in range(20):
        perm = [int(v) for v in rng.permutation(n)]
        ops2 = [q.make_swap_op([perm[0]], [perm[1]]), q.make_swap_op(perm[2:4], perm[4:6]), q.make_swap_op([perm[6], perm[0]], [perm[7], perm[8]]),
                q.make_matrix_op([perm[9]], circuits.H), q.make_swap_op([perm[1]], [perm[10]]),
                q.make_control_op([perm[11]], q.make_swap_op([perm[2]], [perm[3]])), q.make_swap_op([perm[12]], [perm[13]])]
        got2, plan2 = replay(n, ops2, 1 + trial % 2, x)
        assert np.max(np.abs(got2 - O.apply_ops_in_place(n, ops2, x.copy()))) <= 1e-12, perm


@pytest.mark.parametrize("mode", [1 | 4 | 8, 2 | 4 | 8])  # bit 3: keep the relabelled plan even where it is not shorter
@pytest.mark.parametrize("name", ["c2", "qft", "c4", "grover", "fuzz12", "fuzz13", "fuzz14", "c2long", "c2n18", "fuzz17"])
def test_relabelled_tile_plan_replayed_on_cpu_matches_the_oracle(name, mode, sched):
    """option tile_relabel (mode bit 2): the scheduler keeps a logical -> physical map of the qubits, brings the soonest-
    needed ones onto index bits 0..5 with in-tile swaps, turns Swap ops into label exchanges and restores the order with one
    bit-permutation sweep at the end.  Replayed with the numpy model; must still equal the oracle on the ORIGINAL circuit."""
    n = {"fuzz13": 13, "fuzz14": 14, "c2long": 13, "c2n18": 18, "fuzz17": 17}.get(name, 12)
    rng = np.random.default_rng(len(name) * 11 + n)
    ops = {
        "c2": lambda: circuits.h_layer(n) + circuits.c2_random_circuit(n, 120, seed=28),
        "c2long": lambda: circuits.c2_random_circuit(n, 400, seed=3),
        "c2n18": lambda: circuits.c2_random_circuit(n, 160, seed=18),
        "qft": lambda: circuits.c3_qft(n),
        "c4": lambda: circuits.c4_clifford_t(n, 120, seed=32),
        "grover": lambda: circuits.h_layer(n) + circuits.c5_grover_iteration(n),
    }.get(name, lambda: fuzz_circuit(n, rng, 140))()
    x = circuits.random_state(n, seed=n)
    got, plan = replay(n, ops, mode, x)
    assert "circuit" in plan
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, float(np.max(np.abs(want))))
    plain = debug_tile_plan(n, ops, mode & 3)
    if name in ("c2", "c2long", "c4"):
        # (a tile already covers 11 of these 12-13 bits, so nothing can be saved here: at most the closing permutation is added)
        assert len(plan["steps"]) <= len(plain["steps"]) + (2 if sched == 2 else 1), (len(plan["steps"]), len(plain["steps"]))  # (the search shortens the plain plan too)


def test_relabelling_saves_sweeps_at_bench_size():
    """the schedule only (no state): configs[1] / Clifford+T at n = 30 need fewer sweeps with relabelling, the final
    permutation sweep included; every caller op lands in exactly one step or is an absorbed Swap"""
    from rustqip_amd.ops import plan_tiles

    n = 30
    for ops, plain_max, rel_max in ((circuits.c2_random_circuit(n, 256, seed=28), 19, 15), (circuits.c4_clifford_t(n, 256, seed=32), 16, 13),
                                    (circuits.c2_random_circuit(n, 1024, seed=28), 71, 50), (circuits.c3_qft(n), 9, 9)):
        n_plain, rel = len(plan_tiles(n, ops, 1)), plan_tiles(n, ops, 1 | 4)
        assert n_plain <= plain_max and len(rel) <= rel_max, (n_plain, len(rel))
        placed = sorted(i for st in rel for i in st)
        assert len(set(placed)) == len(placed) and all(ops[i].kind == "Swap" for i in set(range(len(ops))) - set(placed))
    # r3: positions claimed by what they buy (the shortest of three plans is kept): the commuting mode with relabelling
    for ops, first_come, searched in ((circuits.c2_random_circuit(n, 256, seed=28), 10, 9), (circuits.c4_clifford_t(n, 256, seed=32), 8, 6),
                                      (circuits.c2_random_circuit(n, 1024, seed=28), 27, 26)):
        q.set_global_option("tile_sched", 0)
        a = len(plan_tiles(n, ops, 2 | 4))
        q.set_global_option("tile_sched", 1)
        b = plan_tiles(n, ops, 2 | 4)
        assert a == first_come and len(b) <= searched, (a, len(b))
        placed = sorted(i for st in b for i in st)
        assert len(set(placed)) == len(placed) and all(ops[i].kind == "Swap" for i in set(range(len(ops))) - set(placed))


def test_tile_plan_for_complex64_states():
    """The f32 plan (8-byte amplitudes: swizzle fold S = 5, matrices rounded to f32) replayed in f64: same
    structure, agreement with the f64 oracle to f32 rounding."""
    from rustqip_amd import _ffi

    n = 12
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 100, seed=5) + circuits.c3_qft(n)[:40]
    x = circuits.random_state(n, seed=1)
    got, plan = replay(n, ops, 1, x, _ffi.QIP_C32)
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) < 1e-5
    assert len(plan["steps"]) < len(ops) / 4


def test_run_time_compiled_segments_build_without_a_gpu():
    """option tile_jit: every multi-gate segment is written out as straight-line HIP source (one pass_* helper call per
    gate, descriptors as constexpr values) and compiled with hiprtc.  hiprtc cross-compiles for gfx950 without a device,
    so the generator is exercised here for every gate shape a segment can hold, in both precisions; that the compiled
    kernels compute the interpreter's results bit for bit is the GPU test's job (tests/test_gpu_f4_tiles.py)."""
    from rustqip_amd import _ffi
    from rustqip_amd.ops import debug_tile_jit

    n = 13
    rng = np.random.default_rng(13)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 60, seed=5) + circuits.c3_qft(n)[:60]
    for _ in range(40):  # the shapes the circuits above lack: swaps, dense 2-qubit gates, controls of every kind
        perm = [int(v) for v in rng.permutation(n)]
        kind = int(rng.integers(0, 6))
        if kind == 5:
            a = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
            g = q.make_matrix_op(perm[:3], np.linalg.qr(a)[0].ravel())
            ops.append(g if rng.integers(0, 2) else q.make_control_op([perm[3]], g))
        elif kind == 0:
            ops.append(q.make_swap_op([perm[0]], [perm[1]]))
        elif kind == 1:
            a = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
            ops.append(q.make_matrix_op(perm[:2], np.linalg.qr(a)[0].ravel()))
        elif kind == 2:
            ops.append(q.make_control_op(perm[:3], q.make_matrix_op([perm[3]], [0.3 + 0.1j, -0.7j, 0.2, 0.9 - 0.4j])))
        elif kind == 3:
            ops.append(q.make_control_op([perm[0]], q.make_swap_op([perm[1]], [perm[2]])))
        else:
            ops.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], [1, 0, 0, cmath.rect(1, 0.7)])))
    for dtype in (_ffi.QIP_C64, _ffi.QIP_C32):
        r = debug_tile_jit(n, ops, 1, dtype)
        assert r["segments"] >= 3 and r["code_bytes"] > 0
        src = r["first_source"]
        assert "constexpr TileGate<T> g" in src and "pass_" in src and "#include \"qip_kernels.h\"" in src
        assert r["all_sources_contain"]("pass_dense3<T") if "all_sources_contain" in r else True
        assert ("typedef double T;" in src) == (dtype == _ffi.QIP_C64)


def test_parametrised_segments_keep_their_source_when_angles_change():
    """option tile_jit = 2 (host hook: mode bit 6): a segment's numbers are kernel data (P[k] reads), its structure is code.
    New rotation angles give the SAME source — the cache key — so a variational loop compiles once; a value that becomes
    exactly 0 or +-1 is structure (zero-skipping, unit entries) and changes it.  Also in the contraction flavour (bit 5)."""
    import re

    from rustqip_amd.ops import debug_tile_jit

    n = 14

    def ansatz(thetas):
        ops = []
        for layer in range(2):
            for t in range(8):
                ops.append(q.make_matrix_op([t], circuits.rz(thetas[layer][t])))
                c, s = math.cos(thetas[layer][t] / 2), math.sin(thetas[layer][t] / 2)
                ops.append(q.make_matrix_op([t + 3], [c, -s, s, c]))  # a real rotation
                ops.append(q.make_control_op([t], q.make_matrix_op([t + 5], circuits.X)))
            ops.append(q.make_control_op([1], q.make_matrix_op([9], [1, 0, 0, cmath.rect(1, thetas[layer][0])])))
        return ops

    rng = np.random.default_rng(1)
    a = debug_tile_jit(n, ansatz(rng.uniform(0.1, 3, (2, 8))), 1 | 64)
    b = debug_tile_jit(n, ansatz(rng.uniform(0.1, 3, (2, 8))), 1 | 64)
    assert a["segments"] >= 1 and a["first_source"] == b["first_source"] and a["source_bytes"] == b["source_bytes"]
    src = a["first_source"]
    assert "const T* __restrict__ P" in src and len(re.findall(r"P\[\d+\]", src)) >= 16 and "constexpr TileGate<T> g" not in src
    assert "{0.0, 1.0}" in src or "{1.0, 0.0}" in src  # units and zeros stay literals (X, the unit diagonal entry)
    lit_a = debug_tile_jit(n, ansatz(rng.uniform(0.1, 3, (2, 8))), 1)
    lit_b = debug_tile_jit(n, ansatz(rng.uniform(0.1, 3, (2, 8))), 1)
    assert lit_a["first_source"] != lit_b["first_source"] and "P[" not in lit_a["first_source"]
    # an angle of exactly 0 turns Rz into the identity on one side and the rotation into units / zeros: different structure
    th = rng.uniform(0.1, 3, (2, 8))
    th[0][2] = 0.0
    c = debug_tile_jit(n, ansatz(th), 1 | 64)
    assert c["first_source"] != a["first_source"]
    d = debug_tile_jit(n, ansatz(rng.uniform(0.1, 3, (2, 8))), 2 | 32 | 64)  # tile = 2, contraction allowed, parametrised
    assert d["segments"] >= 1 and d["code_bytes"] > 0


def test_sliced_and_packed_store_variants_of_segments_compile():
    """r5: the variants of a run-time-compiled segment that the sharded state's overlapped exchange launches — the sweep in parts
    (an extra opened position + the part's bits as a kernel argument) with the remap's gather riding in the store (dst_of) — for
    11-bit and wide tiles, both precisions: generated and compiled for gfx950 here, run on the GPU by tests/dist_worker_gpu.py."""
    from rustqip_amd.ops import debug_tile_jit

    n = 24  # (room for two positions above 11 outside a 13-bit tile)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 60, seed=5)
    for dtype in (_ffi.QIP_C64, _ffi.QIP_C32):
        for mode in (1 | 64, 1 | 16 | 64 | 256, 2 | 16 | 32 | 64 | 128):
            plain = debug_tile_jit(n, ops, mode, dtype)
            var = debug_tile_jit(n, ops, mode | 2048, dtype)
            src = var["first_source"]
            src = src if isinstance(src, str) else src.decode()
            assert var["segments"] == plain["segments"] and var["code_bytes"] > 0
            assert "slice_or" in src and "dst_of(" in src and "A* __restrict__ out" in src
            assert "slice_or" not in (plain["first_source"] if isinstance(plain["first_source"], str) else plain["first_source"].decode())


def test_merged_diagonal_runs_are_generated_and_compile():
    """option tile_merge (host hook: mode bit 7, with tile = 2 and contraction bit 5): a run of >= 3 consecutive diagonal gates
    becomes products of factors per element set — outside-the-tile controls stay uniform branches, lane-bit conditions selects —
    and the source compiles for gfx950 in both precisions, also with the numbers as kernel data."""
    from rustqip_amd import _ffi
    from rustqip_amd.ops import debug_tile_jit

    n = 20
    for dtype in (_ffi.QIP_C64, _ffi.QIP_C32):
        r = debug_tile_jit(n, circuits.c3_qft(n), 2 | 32 | 128, dtype)
        src = r["first_source"]
        assert r["segments"] >= 2 and r["code_bytes"] > 0
        assert "one run of diagonal gates" in src and "A F0 = " in src and "F1 = cmul(F1, " in src
        assert "QIP_KEEP_BRANCH(); F" in src          # a control outside the tile: a branch around the product
        assert "tile_sel(((tb & " in src               # a control on a lane bit: entry or 1 per lane
        plain = debug_tile_jit(n, circuits.c3_qft(n), 2 | 32, dtype)
        assert "one run of diagonal gates" not in plain["first_source"]
        assert src.count("cmul(") < plain["first_source"].count("pass_scale") * 2  # far fewer products than gates x elements
        par = debug_tile_jit(n, circuits.c3_qft(n), 2 | 32 | 64 | 128, dtype)
        assert "one run of diagonal gates" in par["first_source"] and "P[" in par["first_source"] and par["code_bytes"] > 0


@pytest.mark.parametrize("mode", [1, 1 | 4 | 8])
@pytest.mark.parametrize("name", ["c2", "c4", "grover", "qft", "fuzz14", "fuzz17"])
def test_tile_1_plans_only_reorder_what_commutes_exactly(name, mode, sched):
    """`tile` = 1 promises IEEE equality with the gate-by-gate path.  The scheduler hoists gates between segments and (r3) orders
    the gates inside a segment for the fewest LDS passes — both only across gates that commute with one of the two rounding-free.
    Checked with the oracle's own arithmetic: the circuit applied in the plan's order equals the circuit order BIT FOR BIT."""
    import dataclasses

    n = {"fuzz14": 14, "fuzz17": 17}.get(name, 13)
    rng = np.random.default_rng(len(name) * 13 + n)
    ops = {
        "c2": lambda: circuits.h_layer(n) + circuits.c2_random_circuit(n, 200, seed=28),
        "c4": lambda: circuits.h_layer(n) + circuits.c4_clifford_t(n, 200, seed=32),
        "grover": lambda: circuits.h_layer(n) + circuits.c5_grover_iteration(n),
        "qft": lambda: circuits.c3_qft(n),
    }.get(name, lambda: fuzz_circuit(n, rng, 160))()
    plan = debug_tile_plan(n, ops, mode)
    circuit = ops
    if "circuit" in plan:  # relabelled: the caller's ops under the labels in force + inserted swaps; absorbed swaps are label exchanges
        circuit = [q.make_swap_op(c["i"][:1], c["i"][1:]) if c["o"] < 0 else dataclasses.replace(ops[c["o"]], indices=list(c["i"])) for c in plan["circuit"]]
    in_plan_order, moved = [], 0
    for step in plan["steps"]:
        if "perm" in step and not step["ops"]:
            continue  # (the restoring sweep: compared below through the final layout instead)
        idx = step["ops"]
        if "order" in step:
            assert sorted(step["order"]) == list(range(len(idx)))
            moved += sum(1 for k, o in enumerate(step["order"]) if k != o)
            idx = [idx[o] for o in step["order"]]
        in_plan_order += [circuit[i] for i in idx]
    x = circuits.random_state(n, seed=n + 1)
    got = O.apply_ops_in_place(n, in_plan_order, x.copy())
    if "circuit" in plan:
        # the relabelled circuit leaves the qubits permuted: undo with the plan's closing permutation, if any
        last = plan["steps"][-1]
        if "perm" in last and not last["ops"]:
            j = np.arange(1 << n, dtype=np.uint64)
            src = np.zeros_like(j)
            for dbit, sbit in enumerate(last["perm"]):
                src |= ((j >> np.uint64(dbit)) & np.uint64(1)) << np.uint64(sbit)
            got = got[src.astype(np.int64)]
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.array_equal(got.view(np.float64), want.view(np.float64)) or np.array_equal(got, want)
    if sched == 2 and name in ("c2", "fuzz17"):
        assert moved > 0  # the reordering really happened


def test_wide_tile_segments_plan_and_compile_without_a_gpu():
    """r4, option tile_wide (mode bit 4 of the host hooks): segments over a 13-bit register-resident tile claim seven free
    positions — fewer sweeps for the same circuit, every op still in exactly one step — and their generated source (32-element
    register arrays, LDS transpositions in four quarters) compiles with hiprtc for gfx950 without a device, also in the
    parametrised form and with contraction allowed.  (What the compiled kernels compute is checked on the GPU against the
    narrow sweeps bit for bit and against the oracle: tests/test_gpu_f4_tiles.py::test_wide_tiles_...)"""
    from rustqip_amd.ops import debug_tile_jit, plan_tiles

    n = 30
    for ops, narrow_max, wide_max in ((circuits.c2_random_circuit(n, 256, seed=28), 18, 13), (circuits.c4_clifford_t(n, 256, seed=32), 15, 11),
                                      (circuits.c5_grover_iteration(n), 15, 11)):
        a, b = plan_tiles(n, ops, 1), plan_tiles(n, ops, 1 | 16)
        assert len(a) <= narrow_max and len(b) <= wide_max, (len(a), len(b))
        assert sorted(i for st in b for i in st) == list(range(len(ops)))
    assert len(plan_tiles(n, circuits.c2_random_circuit(n, 256, seed=28), 1 | 4 | 16)) <= 10
    n = 16
    rng = np.random.default_rng(4)
    u2, u3 = rand_unitary(2, rng), rand_unitary(3, rng)
    ops = circuits.c2_random_circuit(n, 60, seed=9) + [q.make_matrix_op([3, 12], u2.ravel()), q.make_matrix_op([15, 0, 7], u3.ravel()),
                                                        q.make_control_op([1, 14], q.make_matrix_op([9], circuits.H)), q.make_swap_op([2], [13])]
    # (bit 8: register pins after block-uniform branches; bit 9: dense 3-qubit gates written out group by group — literal and parametrised)
    for mode in (1 | 16, 1 | 16 | 64, 2 | 16 | 32 | 64, 1 | 4 | 8 | 16 | 64, 2 | 16 | 32 | 64 | 128, 1 | 16 | 64 | 256, 1 | 16 | 512, 2 | 16 | 32 | 64 | 256 | 512):
        r = debug_tile_jit(n, ops, mode)
        assert r["segments"] >= 1 and r["code_bytes"] > 0, (mode, r)
        src = r["first_source"] if isinstance(r["first_source"], str) else r["first_source"].decode()
        assert "A e0[32];" in src and "__launch_bounds__(256, 2)" in src
    from rustqip_amd import _ffi

    for mode in (1 | 16 | 64, 1 | 16 | 64 | 256, 1 | 16 | 64 | 256 | 512):
        r = debug_tile_jit(n, ops, mode, _ffi.QIP_C32)
        assert r["segments"] >= 1


@pytest.mark.parametrize("mode", [1 | 16, 2 | 16, 1 | 4 | 8 | 16, 2 | 4 | 16])
@pytest.mark.parametrize("name", ["c2", "fuzz", "qft", "grover_k3"])
def test_wide_tile_plan_replayed_on_cpu_matches_the_oracle(name, mode):
    """r4: the host half of the wide tiles without a GPU — qip_hip_debug_tile_plan with mode bit 4 exports every wide segment
    (arrangements with their register bits, lane maps, quarter bits and buffer layouts; the gates in 13-bit tile-index space,
    in the order they are applied); `emulate_wide_segment` checks the plan's structure and replays its gates with numpy, and
    the result must match the oracle applying the caller's circuit gate by gate.  (The generated code itself — transpositions
    included — is compiled here with hiprtc and compared bit for bit with the 11-bit sweeps on the GPU.)"""
    n = 15
    rng = np.random.default_rng(15)
    ops = {"c2": circuits.h_layer(n) + circuits.c2_random_circuit(n, 120, seed=28), "fuzz": fuzz_circuit(n, rng, 120),
           "qft": circuits.c3_qft(n), "grover_k3": circuits.c5_grover_iteration(n, dense_k3=True)}[name]
    x = circuits.random_state(n, seed=n)
    got, plan = replay(n, ops, mode, x)
    assert any(st.get("wide") for st in plan["steps"])
    want = O.apply_ops_in_place(n, ops, x.copy())
    assert np.max(np.abs(got - want)) <= 1e-12, (name, mode)


# ---- k_sparse_tile (r4): the host half of the in-place SparseMatrix kernel without a GPU ------------------------------------------
def debug_sparse_tile(n, op, dtype=None):
    from rustqip_amd import _ffi
    from rustqip_amd.ops import debug_sparse_tile as hook

    return hook(n, op, _ffi.QIP_C64 if dtype is None else dtype)


def emulate_sparse_tile(state, n, plan):
    """k_sparse_tile on a numpy vector, restated from the kernel: block -> tile base (tile_block_base: the block number spread over
    the positions `ins` leaves open, in the space where position 5 and p5 have traded places, the controls outside the tile read 1),
    lane -> the six row positions (0..4 and p5), tile row -> the op's other positions; every output row folds its stored entries
    in stored order from 0 out of the staged tile; lanes whose in-row controls do not all read 1 keep their amplitudes."""
    kh, p5, E = plan["kh"], plan["p5"], plan["E"]
    hpos, low_op, low_ctl, nlow = plan["hpos"], plan["low_op"], plan["low_ctl"], plan["nlow"]
    nnz = np.array(plan["nnz"], dtype=np.int64)
    slot = np.array(plan["slot"], dtype=np.int64).reshape(-1, E)
    val = np.array([complex(a, b) for a, b in plan["val"]], dtype=state.dtype).reshape(-1, E)
    assert plan["threads"] == 1 << (kh + 3) and len(hpos) == kh and len(nnz) == 1 << (kh + nlow)
    ins_pos, ormask = plan["ins_pos"], plan["ins_ormask"]
    assert ins_pos == sorted(ins_pos)
    lane = np.arange(64, dtype=np.int64)
    lane_off = (lane & 31) | ((lane >> 5) << p5)
    rows = np.arange(1 << kh, dtype=np.int64)
    row_off = np.zeros_like(rows)
    for j, hp in enumerate(hpos):
        row_off |= ((rows >> j) & 1) << hp
    ml = np.zeros(64, dtype=np.int64)  # the op's lane bits, packed
    o = 0
    for b in range(6):
        if (low_op >> b) & 1:
            ml |= ((lane >> b) & 1) << o
            o += 1
    assert o == nlow
    keep = lane & ~low_op
    active = (lane & low_ctl) == low_ctl
    out = state.copy()
    touched = np.zeros(1 << n, dtype=bool)
    for blk in range(plan["ntiles"]):
        w = blk << 6
        for p in ins_pos:  # insert_bits: a zero at every listed position, ascending, then the ones
            w = ((w >> p) << (p + 1)) | (w & ((1 << p) - 1))
        w |= ormask
        if p5 != 5:  # back from the traded space: the bit that stands at p5 belongs at 5
            b = (w >> p5) & 1
            w = (w & ~(1 << p5)) | (b << 5)
        g = (w | row_off[:, None] | lane_off[None, :])  # (rows, lanes) global indices
        assert not touched[g].any()
        touched[g] = True
        tile = state[g].reshape(-1)  # tile index = (row << 6) | lane
        m = ml[None, :] | (rows[:, None] << nlow)
        acc = np.zeros(g.shape, dtype=state.dtype)
        for e in range(E):
            v, xx = val[m, e], tile[slot[m, e] | keep[None, :]]
            # num-complex's product, component by component (numpy's own complex multiply may fuse): cmul in qip_kernels.h
            term = (v.real * xx.real - v.imag * xx.imag) + 1j * (v.real * xx.imag + v.imag * xx.real)
            acc = np.where(e < nnz[m], acc + term.astype(state.dtype), acc)
        out[g] = np.where(active[None, :], acc, state[g])
    return out, touched


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_sparse_tile_plan_replayed_on_cpu_matches_the_oracle(dtype):
    """r4: qip_hip_debug_sparse_tile exports what the host ships to k_sparse_tile (descriptor, block-base insert list, row table);
    the numpy model above replays it and must reproduce the oracle bit for bit (same fold order) — op positions inside and outside
    the wave row, the split position as an op bit and as a control, position 5, controls inside / outside, ragged rows, 1 / 2 / 4
    entries per row, seven positions outside the row; and the shapes that must NOT take this kernel."""
    from rustqip_amd import _ffi

    n = 16
    f64 = dtype == np.complex128
    code = _ffi.QIP_C64 if f64 else _ffi.QIP_C32
    rng = np.random.default_rng(16)
    x = circuits.random_state(n, seed=4, dtype=dtype)

    def rand_rows(k, width):
        return [[(int(rng.integers(0, 1 << k)), complex(rng.standard_normal(), rng.standard_normal())) for _ in range(int(rng.integers(1, width + 1)))]
                for _ in range(1 << k)]

    perm = rng.permutation(1 << 8)
    cases = {
        "perm_phase8_kh7": q.make_sparse_matrix_op([15, 0, 9, 3, 11 if f64 else 12, 5, 1, 2], [[(int(perm[r]), complex(np.exp(0.1j * r)))] for r in range(256)]),
        "two_per_row6": q.make_sparse_matrix_op([2, 15, 8, 0, 9, 5], rand_rows(6, 2)),
        "four_per_row7_two_in_row": q.make_sparse_matrix_op([4, 1, 14, 9, 13, 0, 7], rand_rows(7, 4)),
        "split_position_op_and_position5": q.make_sparse_matrix_op([4, 0, 10, 3, 9, 1], rand_rows(6, 3)),
        "controls_in_and_out": q.make_control_op([13, 1, 11], q.make_sparse_matrix_op([0, 2, 4, 8, 6, 15], rand_rows(6, 2))),
        "split_position_ctl": q.make_control_op([4, 10], q.make_sparse_matrix_op([0, 2, 5, 8, 7, 15], rand_rows(6, 2))),
        "k5": q.make_sparse_matrix_op([0, 3, 6, 9, 1], rand_rows(5, 2)),
    }
    for name, op in cases.items():
        plan = debug_sparse_tile(n, op, code)
        if name == "k5" and not f64:
            assert plan["applies"] == 0  # (Complex<f32>: k = 4, 5 stay with one group per lane)
            continue
        assert plan["applies"] == 1, name
        got, touched = emulate_sparse_tile(x, n, plan)
        want = O.apply_ops_in_place(n, [op], x.copy())
        assert np.array_equal(got, want), name
        # every amplitude inside the op's controlled sub-space is written exactly once, nothing outside the outside controls is read
        nctl_out = n - 6 - plan["kh"] - int(math.log2(plan["ntiles"]))
        assert int(touched.sum()) == (1 << n) >> nctl_out, name
    # not this kernel: five entries in a row, only two positions outside the wave row, eight outside, a state too small, a dense op
    for name, op, nn in (("five_per_row", q.make_sparse_matrix_op([2, 15, 8, 0, 9, 5], rand_rows(6, 5) + []), n),
                         ("two_outside", q.make_sparse_matrix_op([15, 14, 13, 12, 11 if not f64 else 4, 0, 1], rand_rows(7, 2)), n),
                         ("eight_outside", q.make_sparse_matrix_op([0, 1, 2, 3, 5, 6, 7, 8], [[(r, 1.0)] for r in range(256)]), n),
                         ("small_state", q.make_sparse_matrix_op([0, 1, 2, 3, 4, 5], rand_rows(6, 2)), 10),
                         ("dense", q.make_matrix_op([0, 9], np.eye(4).ravel()), n)):
        if name == "five_per_row":
            op = q.make_sparse_matrix_op([2, 15, 8, 0, 9, 5], [[(c, 1.0) for c in range(5)] for _ in range(64)])
        assert debug_sparse_tile(nn, op, code)["applies"] == 0, name
