"""TEST INFRASTRUCTURE — oracle comparison at sizes where the full vector cannot be compared.

SURVEY.md §8(d): "at full n, … a window compare where the oracle can be evaluated (apply_op_row on
selected rows is cheap: matrix_ops.rs:38-59)".  A row of `op · x` only reads the columns that differ from the
row on the op's own bit positions (matrix_ops.rs:74-93, sub_to_full :24-30).  So for a window of 2^w
consecutive rows the needed inputs are the <= 2^h windows obtained by setting the op's h bit positions >= w
every possible way: together they form a sub-cube of the index space that is CLOSED under the op.  On that
sub-cube the op acts exactly like the same MatrixOp on an m = w + h qubit register (same matrix rows, same
column order, hence the same sequence of rounded operations per row), which is what the oracle evaluates:

    before  = the 2^h windows downloaded from the device before the gate         (2^m amplitudes)
    want    = qip_oracle apply_op_overwrite(m, op', before)   and, on sampled rows, qip_oracle apply_op_row
    after   = the same windows downloaded after the gate

Bits an op only TESTS — its controls, and targets in which its matrix is diagonal — need not be part of the sub-cube:
outside it such a bit is constant over the whole cube, so the op either is the identity there (a control reads 0), or it
is the same op without that control / with the diagonal block its bit value selects (same matrix rows in the same column
order, so the same rounded operations).  Only the bits an op EXCHANGES amplitudes across must be closed over.  That is
what lets whole QFT segments (5 H + up to 135 controlled phases with controls on every index bit) be checked at n = 30.

Whole-vector guard: sub-cubes prove the compared rows right; they cannot see a stray write elsewhere.  `check_ops` can
therefore keep a TWIN state in lock step — every op applied gate by gate through the literal out-of-place kernel
(option force_generic), an entirely different code path — and compare the two states over all 2^n amplitudes on the device
(HipState.max_abs_diff) after every checked step.

Only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this module; the product never does.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from rustqip_amd.ops import MatrixOp


def _remap(op: MatrixOp, qmap: Dict[int, int]) -> MatrixOp:
    inner = _remap(op.inner, qmap) if op.inner is not None else None
    return MatrixOp(op.kind, [qmap.get(int(q), 0) for q in op.indices], data=op.data, rows=op.rows, half=op.half,
                    n_controls=op.n_controls, inner=inner)


def flatten(op: MatrixOp):
    """(controls, targets, core): nested Control ops flattened as the reference does (ops.rs:150-154); `core` is the
    innermost non-Control op, whose own index list is ignored at apply time (matrix_ops.rs:108)."""
    idx = [int(q) for q in op.indices]
    ctrls: List[int] = []
    cur, at = op, 0
    while cur.kind == "Control":
        ctrls += idx[at:at + cur.n_controls]
        at += cur.n_controls
        cur = cur.inner
    return ctrls, idx[at:], cur


def diagonal_targets(core: MatrixOp, k: int) -> List[bool]:
    """diag[j]: the dense matrix has no entry linking sub-indices that differ in target j (sub-index bit k-1-j)"""
    if core.kind != "Matrix":
        return [False] * k
    m = np.asarray(core.data).reshape(1 << k, 1 << k)
    rows, cols = np.nonzero(m)
    out = []
    for j in range(k):
        bit = k - 1 - j
        out.append(bool(np.all(((rows >> bit) & 1) == ((cols >> bit) & 1))))
    return out


def exchange_positions(n: int, op: MatrixOp) -> List[int]:
    """index bit positions across which `op` moves amplitude (the bits a closed sub-cube must contain)"""
    ctrls, targets, core = flatten(op)
    diag = diagonal_targets(core, len(targets))
    return [n - 1 - q for q, d in zip(targets, diag) if not d]


class SubCube:
    """Index bits `low` = 0..w-1 plus the sorted positions `high`; every other bit is fixed by `base`."""

    def __init__(self, n: int, positions: Sequence[int], w_max: int = 16, m_max: int = 22):
        pos = sorted(set(int(p) for p in positions))
        w = min(w_max, n)
        while w > 0 and w + sum(1 for p in pos if p >= w) > min(m_max, n):
            w -= 1
        self.n, self.w = n, w
        self.high = [p for p in pos if p >= w]
        self.m = w + len(self.high)
        self.ok = w >= 4 and self.m <= m_max
        self.vmask = ((1 << w) - 1) | sum(1 << p for p in self.high)

    def sub_position(self, p: int) -> int:
        return p if p < self.w else self.w + self.high.index(p)

    def offsets(self, base: int) -> List[int]:
        base &= ~self.vmask & ((1 << self.n) - 1)
        out = []
        for h in range(1 << len(self.high)):
            off = base
            for j, p in enumerate(self.high):
                off |= ((h >> j) & 1) << p
            out.append(off)
        return out

    def gather(self, download: Callable[[int, int], np.ndarray], base: int) -> np.ndarray:
        return np.concatenate([download(off, 1 << self.w) for off in self.offsets(base)])

    def gather_state(self, state, base: int) -> np.ndarray:
        """the sub-cube at `base` out of a state: window by window (a HipState: contiguous copies), or in ONE indexed
        gather when the state offers download_logical (a sharded state, whose logical windows are scattered over the
        ranks' shards: rustqip_amd.sharded.DistState)"""
        if hasattr(state, "download_logical"):
            w = np.arange(1 << self.w, dtype=np.uint64)
            return state.download_logical(np.concatenate([w + np.uint64(off) for off in self.offsets(base)]))
        return self.gather(state.download, base)

    def localize(self, op: MatrixOp) -> MatrixOp:
        qmap = {}
        for q in op.indices:
            qmap[int(q)] = self.m - 1 - self.sub_position(self.n - 1 - int(q))
        return _remap(op, qmap)

    def inside(self, p: int) -> bool:
        return p < self.w or p in self.high

    def localize_at(self, op: MatrixOp, base: int) -> Optional[MatrixOp]:
        """`op` as it acts on the sub-cube at `base` (see the module docstring): None when a control outside the cube
        reads 0 there.  Bits outside the cube must be controls or diagonal targets."""
        ctrls, targets, core = flatten(op)
        if all(self.inside(self.n - 1 - q) for q in ctrls + targets):
            return self.localize(op)
        sub = lambda q: self.m - 1 - self.sub_position(self.n - 1 - q)  # noqa: E731
        bit = lambda q: (base >> (self.n - 1 - q)) & 1                  # noqa: E731
        keep_c = []
        for q in ctrls:
            if self.inside(self.n - 1 - q):
                keep_c.append(sub(q))
            elif bit(q) == 0:
                return None
        k = len(targets)
        fixed = [j for j, q in enumerate(targets) if not self.inside(self.n - 1 - q)]
        if fixed:
            if core.kind != "Matrix":
                raise ValueError("a %s op with a target outside the sub-cube" % core.kind)
            diag = diagonal_targets(core, k)
            if not all(diag[j] for j in fixed):
                raise ValueError("an exchanging target outside the sub-cube")
            m = np.asarray(core.data).reshape(1 << k, 1 << k)
            keep = [j for j in range(k) if j not in fixed]
            sel = []
            for r in range(1 << len(keep)):
                full = 0
                for jj, j in enumerate(keep):
                    full |= ((r >> (len(keep) - 1 - jj)) & 1) << (k - 1 - j)
                for j in fixed:
                    full |= bit(targets[j]) << (k - 1 - j)
                sel.append(full)
            block = m[np.ix_(sel, sel)]
            if keep:
                new_targets = [sub(targets[j]) for j in keep]
            else:  # a scalar: the same product as a one-qubit diag(s, s) on a cube qubit the op does not otherwise use
                spare = next(q for q in range(self.m) if q not in keep_c)
                new_targets = [spare]
                block = np.array([[block[0, 0], 0], [0, block[0, 0]]], dtype=np.complex128)
            new_core = MatrixOp.new_matrix(new_targets, block.ravel())
        else:
            new_targets = [sub(q) for q in targets]
            new_core = MatrixOp(core.kind, new_targets, data=core.data, rows=core.rows, half=core.half)
        if not keep_c:
            return new_core
        return MatrixOp.new_control(keep_c, new_targets, new_core)


def default_bases(n: int, seed: int = 0, count: int = 4) -> List[int]:
    """bottom and top of the index space plus seeded random places in between"""
    rng = np.random.default_rng(seed)
    full = (1 << n) - 1
    bases = [0, full]
    while len(bases) < count:
        bases.append(int(rng.integers(0, 1 << 62)) & full)
    return bases[:count]


class Twin:
    """The whole-vector guard: a second state of the same size that follows the checked state gate by gate through the
    literal out-of-place kernel (option force_generic), compared with it over all 2^n amplitudes on the device."""

    def __init__(self, state, make_state: Callable[[], object]):
        self.state = state
        self.twin = make_state()
        self.twin.set_option("force_generic", 1)
        self.twin.copy_from(state)
        self.worst, self.differ, self.compares = 0.0, 0, 0

    def follow(self, ops: Sequence[MatrixOp]) -> dict:
        self.twin.apply_ops(list(ops))
        worst, differ = self.state.max_abs_diff(self.twin)
        self.worst = max(self.worst, worst)
        self.differ += differ
        self.compares += 1
        return {"max_abs_delta": worst, "amplitudes_not_equal": differ}

    def resync(self) -> None:
        """after a leg that is only held to a tolerance: start the next one from identical states again"""
        self.twin.copy_from(self.state)

    def close(self) -> None:
        self.twin.close()


def check_ops(state, n: int, ops: Sequence[MatrixOp], O, bases: Optional[Sequence[int]] = None,
              apply: Optional[Callable[[], None]] = None, rows_sampled: int = 8, w_max: int = 16,
              twin: Optional[Twin] = None) -> Optional[dict]:
    """Apply `ops` to the device state (state.apply_ops, or `apply()`), and compare >= len(bases) sub-cubes of
    2^m amplitudes against the oracle applied to what the device held before.  Returns None when the ops exchange
    amplitudes across too many high bit positions for a sub-cube of reasonable size (nothing is applied then), else
    {"rows": amplitudes compared, "max_abs_delta": ..., "bit_equal": every compared component IEEE-==,
     "row_calls": rows additionally recomputed with apply_op_row, "m": ..., "windows": ...,
     "whole_vector": the twin comparison over all 2^n amplitudes (when a twin is given)}."""
    ops = list(ops)
    positions = [p for op in ops for p in exchange_positions(n, op)]
    cube = SubCube(n, positions, w_max=w_max)
    if not cube.ok:
        return None
    bases = list(bases) if bases is not None else default_bases(n)
    before = [cube.gather_state(state, b) for b in bases]
    if apply is not None:
        apply()
    else:
        state.apply_ops(ops)
    worst, equal, rows, row_calls, active = 0.0, True, 0, 0, 0
    rng = np.random.default_rng(n)
    # a sharded state: every rank takes part in the (collective) gathers, ONE rank evaluates the oracle — its verdict is
    # what sharded_parity broadcasts (N ranks computing the same comparison would only fight over the host's cores)
    compare_here = getattr(state, "oracle_on_this_rank", True)
    for b, x in zip(bases, before):
        local_ops = [lop for lop in (cube.localize_at(op, b) for op in ops) if lop is not None]
        active += len(local_ops)
        got = cube.gather_state(state, b)
        rows += got.size if not compare_here else 0
        if not compare_here:
            continue
        cur, arena = x.copy(), np.zeros_like(x)
        for lop in local_ops:
            O.apply_op_overwrite(cube.m, lop, cur, arena)
            cur, arena = arena, cur
        d = np.abs(got - cur)
        worst = max(worst, float(d.max()))
        equal = equal and bool(np.array_equal(got, cur))
        rows += got.size
        if len(ops) == 1 and len(local_ops) == 1:
            # the literal per-row entry point of the reference (matrix_ops.rs:38-59) on sampled rows, first and last included
            sample = [0, got.size - 1] + [int(r) for r in rng.integers(0, got.size, size=rows_sampled)]
            for r in sample:
                v = O.apply_op_row(cube.m, local_ops[0], x, r)
                worst = max(worst, abs(complex(got[r]) - v))
                equal = equal and (complex(got[r]) == v)
                row_calls += 1
    out = {"rows": rows, "max_abs_delta": worst, "bit_equal": equal, "row_calls": row_calls, "m": cube.m,
           "windows": len(bases) * (1 << len(cube.high)), "ops_active_on_cubes": active}
    if twin is not None:
        out["whole_vector"] = twin.follow(ops)
    return out


def chunk_by_high_bits(n: int, ops: Sequence[MatrixOp], max_high: int = 6, w: int = 16, max_len: int = 64) -> List[List[MatrixOp]]:
    """Cut a circuit into consecutive chunks whose ops together EXCHANGE amplitudes across at most `max_high` bit positions
    >= w, so each chunk has a closed sub-cube of <= 2^(w + max_high) amplitudes (tested-only bits may lie anywhere)."""
    chunks: List[List[MatrixOp]] = []
    cur: List[MatrixOp] = []
    high: set = set()
    for op in ops:
        mine = {p for p in exchange_positions(n, op) if p >= w}
        if cur and (len(high | mine) > max_high or len(cur) >= max_len):
            chunks.append(cur)
            cur, high = [], set()
        cur.append(op)
        high |= mine
    if cur:
        chunks.append(cur)
    return chunks


def check_circuit(state, n: int, ops: Sequence[MatrixOp], O, gate_by_gate: bool = True, seed: int = 0,
                  bases_per_step: int = 4, twin: Optional[Twin] = None, max_len: int = 64) -> dict:
    """check_ops over a whole circuit: gate by gate, or in chunks applied through state.apply_ops (which is how the
    multi-gate tile sweeps are reached).  Aggregates the per-step results."""
    steps = [[op] for op in ops] if gate_by_gate else chunk_by_high_bits(n, ops, max_len=max_len)
    agg = {"gates": 0, "steps": 0, "rows": 0, "row_calls": 0, "max_abs_delta": 0.0, "bit_equal": True, "skipped": 0,
           "windows": 0}
    if twin is not None:
        agg.update({"whole_vector_compares": 0, "whole_vector_max_abs_delta": 0.0, "whole_vector_amplitudes_not_equal": 0})
    for i, chunk in enumerate(steps):
        r = check_ops(state, n, chunk, O, bases=default_bases(n, seed + i, bases_per_step), twin=twin)
        if r is None:
            state.apply_ops(chunk)
            if twin is not None:
                twin.follow(chunk)
            agg["skipped"] += len(chunk)
            continue
        agg["gates"] += len(chunk)
        agg["steps"] += 1
        agg["rows"] += r["rows"]
        agg["row_calls"] += r["row_calls"]
        agg["windows"] += r["windows"]
        agg["max_abs_delta"] = max(agg["max_abs_delta"], r["max_abs_delta"])
        agg["bit_equal"] = agg["bit_equal"] and r["bit_equal"]
        if twin is not None:
            agg["whole_vector_compares"] += 1
            agg["whole_vector_max_abs_delta"] = max(agg["whole_vector_max_abs_delta"], r["whole_vector"]["max_abs_delta"])
            agg["whole_vector_amplitudes_not_equal"] += r["whole_vector"]["amplitudes_not_equal"]
    return agg


def product_state_ops(n: int, seed: int):
    """Ops that turn |0..0> into a seeded PRODUCT state whose amplitudes are pairwise distinct in modulus and phase
    (H, Rz(theta_t), Ry-like real rotation, Rz(phi_t) per qubit), and the per-qubit 2-vectors for its closed form
    amp(idx) = prod_t v[t][bit_t(idx)].  A uniform state hides index-mapping errors; this one does not."""
    import cmath
    import math

    from rustqip_amd.ops import make_matrix_op

    rng = np.random.default_rng(seed)
    ops, vecs = [], []
    for t in range(n):
        a = float(rng.uniform(0.55, 1.0))  # rotation angle: both components well away from 0
        ph0, ph1 = float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(0, 2 * math.pi))
        m = np.array([[math.cos(a) * cmath.rect(1, ph0), -math.sin(a)], [math.sin(a) * cmath.rect(1, ph1), math.cos(a)]],
                     dtype=np.complex128)
        ops.append(make_matrix_op([t], m.ravel()))
        vecs.append((complex(m[0, 0]), complex(m[1, 0])))
    return ops, vecs


def product_state_window(n: int, vecs, offset: int, length: int) -> np.ndarray:
    """closed form of the product state on [offset, offset+length) (qubit t <-> index bit n-1-t)"""
    idx = np.arange(offset, offset + length, dtype=np.uint64)
    out = np.ones(length, dtype=np.complex128)
    for t, (v0, v1) in enumerate(vecs):
        bit = (idx >> np.uint64(n - 1 - t)) & np.uint64(1)
        out *= np.where(bit == 1, v1, v0)
    return out


class ProductGuard:
    """Whole-vector guard with a CLOSED FORM: while only uncontrolled single-qubit gates act on the seeded product state it
    stays a product state, amp(idx) = prod_t v[t][bit_t(idx)], and so do its marginals: the probability of reading m from
    the qubits S is prod_{t in S} |v[t][m_t]|^2 * prod_{t not in S} (|v[t][0]|^2 + |v[t][1]|^2).  Index sets of <= 14
    qubits that together cover every qubit are measured on the device (2^14 outcome sums over all 2^n amplitudes each) and
    compared with that: moduli are pairwise distinct, so an amplitude that lands anywhere but in its own place, or a stray
    write of any size above ~1e-12 of a marginal, shows in at least one set."""

    def __init__(self, n: int, vecs, k: int = 14):
        self.n = n
        self.v = [np.array([v0, v1], dtype=np.complex128) for v0, v1 in vecs]
        k = min(k, n)
        starts = list(range(0, n - k + 1, max(1, k - 6)))
        if starts[-1] != n - k:
            starts.append(n - k)
        self.sets = [list(range(s, s + k)) for s in starts]
        self.worst_rel = 0.0
        self.checks = 0

    def apply(self, op: MatrixOp) -> None:
        ctrls, targets, core = flatten(op)
        if ctrls or len(targets) != 1 or core.kind != "Matrix":
            raise ValueError("the product-state guard follows uncontrolled single-qubit Matrix ops only")
        self.v[targets[0]] = np.asarray(core.data, dtype=np.complex128).reshape(2, 2) @ self.v[targets[0]]

    def marginal(self, qubits: Sequence[int]) -> np.ndarray:
        """out[m], bit i of m <-> qubits[i] (measure_probs' convention)"""
        out = np.ones(1)
        for tq in reversed(list(qubits)):  # the last listed qubit is the top bit of m
            out = np.kron(out, np.abs(self.v[tq]) ** 2)
        rest = 1.0
        for t in range(self.n):
            if t not in qubits:
                rest *= float(np.sum(np.abs(self.v[t]) ** 2))
        return out * rest

    def check(self, state) -> float:
        worst = 0.0
        for s in self.sets:
            got = state.measure_probs(s)
            want = self.marginal(s)
            worst = max(worst, float(np.max(np.abs(got - want)) / np.max(want)))
        self.worst_rel = max(self.worst_rel, worst)
        self.checks += 1
        return worst



# ---- the sharded state (N > 1) at bench shard size --------------------------------------------------------------------------
def sharded_parity(make_state: Callable[[], object], dist, n: int, O, q, circuits, gates: int = 256, quick: bool = False) -> dict:
    """The N > 1 path against the oracle at the size it is TIMED at (the reference's only provision
    for distribution is the window arguments of apply_op, matrix_ops.rs:74-93,96-97).  `make_state()` -> a
    rustqip_amd.sharded.DistState of n qubits over dist's ranks.  Collective: every rank calls it with the same arguments.

    Two nets, as on one GPU: closed sub-cubes of the LOGICAL index space, gathered from whichever ranks hold them through
    the layout (DistState.download_logical), against the oracle's apply_op_overwrite / apply_op_row; and a whole-vector
    guard — a TWIN sharded state that follows batch by batch with every shard on the literal out-of-place kernel
    (force_generic), compared shard by shard over all 2^n amplitudes after every step, plus closed-form marginals of the
    seeded product state through qip_hip_dist_measure_probs while only single-qubit gates have acted.

    Legs: the headline gate by gate; the configs[1] mix gate by gate and as tile sweeps on the shards; Clifford+T and the
    dense-k3 Grover iteration (tile = 0 and tile = 1); and a circuit built so that the remap HAS to gather index bit 0
    (the k_permute_bits route of the pack).  The exchanges, pack sweeps and routes taken are reported from the handle's
    own counters.  Rank 0 evaluates the oracle; its summary is broadcast, so every rank returns the same dict."""
    import time

    t0 = time.perf_counter()
    rank, world = dist.get_rank(), dist.get_world_size()
    st = make_state()
    st.oracle_on_this_rank = rank == 0
    st.init_basis(0)
    twin = Twin(st, make_state)  # both fresh: same (identity) layout; from here on they see the same batches
    ops0, vecs = product_state_ops(n, seed=n)
    st.apply_ops(ops0)
    prep = twin.follow(ops0)
    init_err = 0.0
    for off in (0, (1 << n) // 3, (1 << n) - (1 << 16)):
        got = st.download(off, 1 << 16)
        want = product_state_window(n, vecs, off, 1 << 16)
        init_err = max(init_err, float(np.max(np.abs(got - want) / np.abs(want))))
    guard = ProductGuard(n, vecs)
    guard.check(st)
    st.comm_stats()
    legs: Dict[str, dict] = {}

    def leg(name, ops, exact, gate_by_gate=False, seed=0, bases=2, max_len=64, **options):
        for k, v in options.items():
            st.set_option(k, v)
        r = check_circuit(st, n, ops, O, gate_by_gate=gate_by_gate, seed=seed, bases_per_step=bases, twin=twin, max_len=max_len)
        for k in options:
            st.set_option(k, 0)
        r["options"] = options
        r["bar"] = "IEEE-equal" if exact else "1e-12"
        r["comm"] = {k: v for k, v in st.comm_stats().items() if k in ("remaps", "pack_sweeps", "packs_via_permute_bits")}
        r["ok"] = bool((r["bit_equal"] and r["whole_vector_amplitudes_not_equal"] == 0) if exact
                       else (r["max_abs_delta"] <= 1e-12 and r["whole_vector_max_abs_delta"] <= 1e-12))
        if not exact:
            twin.resync()
        legs[name] = r

    head = circuits.c2_random_circuit(n, gates, seed=28, single_only=True)
    mixed = circuits.c2_random_circuit(n, gates, seed=28)
    n_gg = 8 if quick else 24
    for k0 in range(0, n_gg, 8):
        leg("single_qubit_gate_by_gate_%d" % (k0 // 8), head[k0:k0 + 8], True, gate_by_gate=True, seed=31 + k0, bases=4)
        for op in head[k0:k0 + 8]:
            guard.apply(op)
        guard.check(st)
    marg = {"checks": guard.checks, "index_sets": len(guard.sets), "max_rel_err_vs_closed_form": guard.worst_rel}
    leg("mixed_gate_by_gate", mixed[:8 if quick else 24], True, gate_by_gate=True, seed=32, bases=4)
    leg("mixed_tile1_chunks", mixed[24:88], True, seed=33, tile=1)
    c4 = circuits.c4_clifford_t(n, gates, seed=32)
    leg("configs3_clifford_t_chunks", c4[:32 if quick else 64], True, seed=34)
    leg("configs3_clifford_t_tile1_chunks", c4[64:128], True, seed=35, tile=1)
    g3 = circuits.c5_grover_iteration(n, dense_k3=True)
    # (a dense 8x8 gate runs on the matrix cores gate by gate and as the unfused register fold inside a sweep: 1e-12 bar)
    leg("configs4_grover_dense_k3_chunks", g3[:48 if quick else 96], False, seed=36, max_len=96)
    leg("configs4_grover_dense_k3_tile1_chunks", g3[96:192], False, seed=37, max_len=96, tile=1)
    # the pack HAS to gather index bit 0: the qubit that lives there becomes the least recently used one (a layer of
    # diagonal gates on every other qubit: no exchange), then a batch opens with an H on a qubit that sits on a rank bit
    # and never touches it: farthest next use (never) + least recently used -> it leaves, from physical position 0
    phys = st.layout()
    L = n - int(round(np.log2(world)))
    qa = n - 1 - phys.index(0)
    glob = [n - 1 - p for p in range(n) if phys[p] >= L]
    leg("pack_from_bit0_prelude", [q.make_matrix_op([t], circuits.rz(0.3 + 0.01 * t)) for t in range(n) if t != qa], True, seed=38, max_len=96)
    others = [t for t in range(n) if t != qa and t not in glob]
    batch = [q.make_matrix_op([glob[0]], circuits.H)] + [q.make_matrix_op([t], circuits.H) for t in others[:4]]
    if world > 1:
        leg("pack_from_bit0", batch, True, seed=39, tile=1)
    comm_total = {}
    for r in legs.values():
        for k, v in r["comm"].items():
            comm_total[k] = comm_total.get(k, 0) + v
    twin.close()
    st.close()
    tot = lambda key: sum(r.get(key, 0) for r in legs.values())  # noqa: E731
    exact_legs = [r for r in legs.values() if r["bar"] == "IEEE-equal"]
    out = {
        "checker": "CPU oracle (oracle/qip_oracle.c apply_op_overwrite + apply_op_row) on closed sub-cubes of the LOGICAL index space gathered "
                   "through the layout from whichever ranks hold them (oracle/window_parity.py sharded_parity); whole-vector guard: a twin "
                   "sharded state with every shard on the literal kernel, compared over all 2^n amplitudes after every step + closed-form "
                   "marginals of the product state through qip_hip_dist_measure_probs",
        "n": n, "world": world, "n_local": L,
        "state": "seeded product state, pairwise distinct amplitudes (closed form checked: max rel err %.1e)" % init_err,
        "prep_whole_vector": prep,
        "gates_checked": tot("gates"), "gates_skipped": tot("skipped"), "rows_checked": tot("rows"), "windows": tot("windows"),
        "apply_op_row_calls": tot("row_calls"),
        "max_abs_delta": max(r["max_abs_delta"] for r in exact_legs),
        "bit_equal": bool(all(r["bit_equal"] for r in exact_legs)),
        "max_abs_delta_1e-12_legs": max([r["max_abs_delta"] for r in legs.values() if r["bar"] != "IEEE-equal"] or [0.0]),
        "whole_vector": {"compares": tot("whole_vector_compares"), "amplitudes_per_compare": 1 << n,
                         "amplitudes_not_equal_in_IEEE_legs": sum(r["whole_vector_amplitudes_not_equal"] for r in exact_legs),
                         "max_abs_delta_all_legs": max(r["whole_vector_max_abs_delta"] for r in legs.values())},
        "product_state_marginals": marg,
        "remaps_exercised": comm_total.get("remaps", 0), "pack_sweeps": comm_total.get("pack_sweeps", 0),
        "packs_via_permute_bits": comm_total.get("packs_via_permute_bits", 0),
        "tolerance": 1e-12,
        "all_legs_ok": bool(all(r["ok"] for r in legs.values()) and prep["amplitudes_not_equal"] == 0 and init_err <= 1e-12
                            and guard.worst_rel <= 1e-11),
        "legs": legs,
        "seconds": round(time.perf_counter() - t0, 2),
    }
    box = [out]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return box[0]
