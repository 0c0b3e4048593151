"""State vector sharded over 2^g GPUs, one process per GPU, RCCL over xGMI.

The reference is single-process; its only provision for distribution is the
input_offset / output_offset window arguments (qip-iterators/src/matrix_ops.rs:96-97,
qip/src/state_ops/measurement_ops.rs:17-19).  This module is the MI355X-native realisation:

  * layout   rank r owns the 2^L (L = n - g) amplitudes whose top g *physical* index bits are r;
             a logical->physical bit permutation is kept on the host;
  * local    gates whose non-diagonal targets are all physically local run as ordinary local ops;
             controls and diagonal targets on rank bits are resolved on the host per rank (skip /
             restrict the matrix), so they never communicate;
  * exchange when a non-diagonal target sits on a rank bit, ALL g rank bits are exchanged with the
             top g local bits in one all-to-all (each rank keeps 1/G of its shard and sends 1/G to
             every peer).  With the top local bits as partners every piece is a contiguous chunk,
             so there is no pack/unpack pass, and the transfer uses all point-to-point xGMI links
             at once (7 x ~153 GB/s) instead of one link for a pairwise half-shard swap.
             Which logical qubits become global is chosen by next-use distance when the circuit is
             known (plan / run_plan), else least-recently-used; at most g local bit-swap sweeps
             bring them to the top positions first.

The per-rank compute is a pluggable `backend` (HipBackend below: HipState over torch-allocated
HBM on torch's current stream, so kernels and collectives are stream-ordered).  Tests inject a
CPU backend to cover the N > 1 logic with gloo.
"""
from __future__ import annotations

import math
import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .ops import CircuitError, MatrixOp, make_control_op, make_matrix_op, make_sparse_matrix_op, make_swap_op


# ---- op analysis (host bookkeeping) ---------------------------------------------------------------
def flatten(op: MatrixOp) -> Tuple[List[int], MatrixOp, List[int]]:
    """(control qubits, innermost op, target qubits) with nested Controls accumulated the way
    sum_for_control_iterator does (ops.rs:150-154); only the OUTER index list is used
    (matrix_ops.rs:108)."""
    if op.kind != "Control":
        return [], op, list(op.indices)
    n_control, inner = op.n_controls, op.inner
    n_op = len(op.indices) - op.n_controls
    while inner.kind == "Control":
        n_control += inner.n_controls
        n_op = len(inner.indices) - inner.n_controls
        inner = inner.inner
    if n_control + n_op != len(op.indices):
        raise CircuitError("Control op index list does not match its controls + inner op indices")
    return list(op.indices[:n_control]), inner, list(op.indices[n_control:])


def _dense_of(inner: MatrixOp, k: int) -> np.ndarray:
    if inner.kind == "Matrix":
        return np.asarray(inner.data, dtype=np.complex128).reshape(1 << k, 1 << k)
    if inner.kind == "SparseMatrix":
        m = np.zeros((1 << k, 1 << k), dtype=np.complex128)
        for r, row in enumerate(inner.rows):
            for c, v in row:
                m[r, c] += v
        return m
    raise CircuitError(inner.kind)


def diagonal_targets(inner: MatrixOp, k: int) -> List[bool]:
    """diag[j] = the op never changes target j's bit (its matrix is block-diagonal in that bit),
    so that target can live on a rank bit without communication."""
    if inner.kind == "Swap":
        return [False] * k
    if inner.kind == "Matrix" and k > 10:
        return [False] * k
    if inner.kind == "SparseMatrix":
        ok = [True] * k
        for r, row in enumerate(inner.rows):
            for c, v in row:
                if v != 0:
                    diff = r ^ c
                    for j in range(k):
                        if (diff >> (k - 1 - j)) & 1:
                            ok[j] = False
        return ok
    m = _dense_of(inner, k)
    rows, cols = np.nonzero(m)
    diff = np.bitwise_or.reduce(rows ^ cols) if rows.size else 0
    return [not ((int(diff) >> (k - 1 - j)) & 1) for j in range(k)]


class OpInfo:
    __slots__ = ("op", "ctrl", "inner", "tgt", "diag", "nondiag_bits")

    def __init__(self, n: int, op: MatrixOp):
        self.op = op
        self.ctrl, self.inner, self.tgt = flatten(op)
        for qb in self.ctrl + self.tgt:
            if not 0 <= qb < n:
                raise CircuitError(f"qubit index {qb} out of range for n = {n}")
        if len(set(self.ctrl + self.tgt)) != len(self.ctrl + self.tgt):
            raise CircuitError("sharded states need distinct qubit indices in an op")
        self.diag = diagonal_targets(self.inner, len(self.tgt))
        # logical bit positions (n-1-q) that must be physically local
        self.nondiag_bits = [n - 1 - qb for qb, d in zip(self.tgt, self.diag) if not d]


# ---- per-rank compute backends -----------------------------------------------------------------------
class HipBackend:
    """2^L amplitudes in HBM allocated by torch (two buffers: current + exchange target), driven by
    HipState on torch's current stream."""

    def __init__(self, n_local: int, device: int, host_staged_exchange: bool = False, tile: int = 0):
        import torch

        from .state import HipState

        self.torch = torch
        # host_staged_exchange: run the all-to-all on CPU copies (gloo).  Only for exercising this
        # backend with several ranks on ONE GPU (tests); the product path exchanges device buffers
        # over RCCL.
        self.host_staged = host_staged_exchange
        self.n_local = n_local
        self.dev = torch.device("cuda", device)
        N = 1 << n_local
        self.bufs = [torch.zeros(N, dtype=torch.complex128, device=self.dev) for _ in range(2)]
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.state = HipState(n_local, np.complex128, device, wrap_ptr=self.bufs[0].data_ptr(),
                              scratch_ptr=self.bufs[1].data_ptr(), stream=stream)
        if tile:
            self.state.set_option("tile", tile)  # applies to apply_ops batches (ShardedState.run_plan(batched=True))
        self._events: List[tuple] = []

    def _cur(self) -> int:
        return 0 if self.state.device_ptr() == self.bufs[0].data_ptr() else 1

    def apply_op(self, op: MatrixOp) -> None:
        self.state.apply_op(op)

    def apply_ops(self, ops: Sequence[MatrixOp]) -> None:
        self.state.apply_ops(ops)

    def exchange_buffers(self):
        """(send, recv) as real views; after the collective call adopt_recv()."""
        c = self._cur()
        return self.torch.view_as_real(self.bufs[c]), self.torch.view_as_real(self.bufs[1 - c])

    def adopt_recv(self) -> None:
        self.state.swap_buffers()  # the all-to-all wrote the scratch buffer (builder.rs:514 analogue)

    def set_profile(self, v: int) -> None:
        self.state.set_option("profile", int(v))

    def take_profile(self) -> dict:
        prof = self.state.profile()
        self.state.profile_reset()
        return prof

    def all_to_all(self, dist, recv, send) -> None:
        if self.host_staged:
            h_send = send.cpu()
            h_recv = self.torch.empty_like(h_send)
            dist.all_to_all_single(h_recv, h_send)
            recv.copy_(h_recv)
        else:
            dist.all_to_all_single(recv, send)

    def timed_collective(self, fn):
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self._events.append((e0, e1))
        if len(self._events) > 4096:  # nobody is draining: keep the list bounded
            self.collective_ms()

    def collective_ms(self) -> float:
        self.torch.cuda.synchronize(self.dev)
        t = sum(a.elapsed_time(b) for a, b in self._events)
        self._events.clear()
        return t

    def init_basis(self, index: Optional[int]) -> None:
        if index is None:
            self.bufs[self._cur()].zero_()
        else:
            self.state.init_basis(index)

    def upload(self, x: np.ndarray) -> None:
        self.state.upload(x)

    def download(self) -> np.ndarray:
        return self.state.download()

    def norm_sqr(self) -> float:
        return self.state.norm_sqr()

    def measure_probs(self, local_qubits: Sequence[int]) -> np.ndarray:
        return self.state.measure_probs(local_qubits)

    def measure_state(self, local_qubits: Sequence[int], measured: int, prob: float) -> None:
        self.state.measure_state(local_qubits, measured, prob)

    def sync(self) -> None:
        self.state.sync()
        self.torch.cuda.synchronize(self.dev)

    def reduce_tensor(self, arr: np.ndarray):
        dev = "cpu" if self.host_staged else self.dev
        return self.torch.as_tensor(arr, dtype=self.torch.float64, device=dev)


# ---- the sharded state ------------------------------------------------------------------------------------
class ShardedState:
    def __init__(self, n: int, dist, device: int = 0, backend=None):
        self.n = n
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.g = int(math.log2(self.world))
        if 1 << self.g != self.world:
            raise CircuitError("world size must be a power of two")
        self.L = n - self.g
        if self.L < max(self.g, 1):
            raise CircuitError(f"n = {n} is too small to shard over {self.world} ranks")
        self.backend = backend if backend is not None else HipBackend(self.L, device)
        self.phys = list(range(n))  # phys[p] = physical bit position of logical bit position p = n-1-q
        self.last_use = [0] * n
        self.clock = 0
        self.stats = {"remaps": 0, "local_swaps": 0, "bytes_sent_per_rank": 0, "collective_ms": 0.0}

    # -- mapping helpers ------------------------------------------------------------------------------
    def _logical_at(self, pp: int) -> int:
        return self.phys.index(pp)

    def _rank_bit(self, pp: int) -> int:
        return (self.rank >> (pp - self.L)) & 1

    def _local_qubit(self, pp: int) -> int:
        return self.L - 1 - pp  # local qubit index of local physical bit pp

    def physical_index(self, logical_index: int) -> int:
        P = 0
        for p in range(self.n):
            P |= ((logical_index >> p) & 1) << self.phys[p]
        return P

    def logical_indices_of_shard(self) -> np.ndarray:
        """logical index of every local amplitude, in local order."""
        loc = np.arange(1 << self.L, dtype=np.uint64)
        P = loc | (np.uint64(self.rank) << np.uint64(self.L))
        out = np.zeros_like(P)
        for p in range(self.n):
            out |= ((P >> np.uint64(self.phys[p])) & np.uint64(1)) << np.uint64(p)
        return out

    # -- data in / out -----------------------------------------------------------------------------------
    def init_basis(self, logical_index: int) -> None:
        P = self.physical_index(logical_index)
        owner, local = P >> self.L, P & ((1 << self.L) - 1)
        self.backend.init_basis(local if owner == self.rank else None)

    def upload_global(self, x: np.ndarray) -> None:
        """every rank passes the same full logical vector (tests / small n)."""
        self.backend.upload(np.ascontiguousarray(x[self.logical_indices_of_shard()]))

    def download_global(self) -> np.ndarray:
        """full logical vector on every rank (tests / small n)."""
        import torch

        mine = self.backend.download()
        idx = self.logical_indices_of_shard().astype(np.int64)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, (idx, mine))
        out = np.zeros(1 << self.n, dtype=np.complex128)
        for i, v in parts:
            out[i] = v
        return out

    # -- the exchange ---------------------------------------------------------------------------------------
    def _remap(self, must_be_local: Sequence[int], next_use: Optional[Dict[int, int]] = None) -> None:
        g, L = self.g, self.L
        must = set(must_be_local)
        cand = [p for p in range(self.n) if self.phys[p] < L and p not in must]
        if len(cand) < g:
            raise CircuitError("op touches too many qubits to keep local on this shard size")
        if next_use is not None:
            cand.sort(key=lambda p: (-next_use.get(p, 1 << 60), self.last_use[p]))
        else:
            cand.sort(key=lambda p: self.last_use[p])
        new_globals = cand[:g]
        # bring them to the top-g local physical positions with local bit swaps
        top = list(range(L - g, L))
        sitting = [p for p in new_globals if self.phys[p] in top]
        free_top = [t for t in top if self._logical_at(t) not in new_globals]
        for p in new_globals:
            if p in sitting:
                continue
            t = free_top.pop()
            other = self._logical_at(t)
            self.backend.apply_op(make_swap_op([self._local_qubit(self.phys[p])], [self._local_qubit(t)]))
            self.phys[other], self.phys[p] = self.phys[p], t
            self.stats["local_swaps"] += 1
        # one all-to-all: chunk c of rank r  ->  chunk r of rank c
        send, recv = self.backend.exchange_buffers()
        self.backend.timed_collective(lambda: self.backend.all_to_all(self.dist, recv, send))
        self.backend.adopt_recv()
        for j in range(g):
            a, b = self._logical_at(L - g + j), self._logical_at(L + j)
            self.phys[a], self.phys[b] = L + j, L - g + j
        self.stats["remaps"] += 1
        self.stats["bytes_sent_per_rank"] += (16 << L) * (self.world - 1) // self.world

    # -- gates --------------------------------------------------------------------------------------------------
    def _localize(self, info: OpInfo) -> Optional[MatrixOp]:
        """The op this rank runs on its shard (local qubit indices), or None when it is the identity here."""
        n, L = self.n, self.L
        local_ctrl: List[int] = []
        for qb in info.ctrl:
            pp = self.phys[n - 1 - qb]
            if pp >= L:
                if self._rank_bit(pp) == 0:
                    return None
            else:
                local_ctrl.append(self._local_qubit(pp))
        k = len(info.tgt)
        tpp = [self.phys[n - 1 - qb] for qb in info.tgt]
        glob = [j for j in range(k) if tpp[j] >= L]
        if not glob:
            inner = info.inner
            ltgt = [self._local_qubit(pp) for pp in tpp]
            if inner.kind == "Matrix":
                loc = MatrixOp("Matrix", ltgt, data=inner.data)
            elif inner.kind == "SparseMatrix":
                loc = MatrixOp("SparseMatrix", ltgt, rows=inner.rows)
            else:
                loc = MatrixOp("Swap", ltgt, half=inner.half)
        else:
            # diagonal targets on rank bits: keep the block selected by this rank's bits
            m = _dense_of(info.inner, k)
            keep = [j for j in range(k) if j not in glob]
            sel = []
            for s in range(1 << len(keep)):
                full = 0
                for j in range(k):
                    bit = self._rank_bit(tpp[j]) if j in glob else (s >> (len(keep) - 1 - keep.index(j))) & 1
                    full |= bit << (k - 1 - j)
                sel.append(full)
            sub = m[np.ix_(sel, sel)]
            if keep:
                loc = make_matrix_op([self._local_qubit(tpp[j]) for j in keep], sub.ravel())
            else:
                d = complex(sub[0, 0])
                if d == 1:
                    return None
                if local_ctrl:
                    last = local_ctrl.pop()
                    loc = make_matrix_op([last], [1, 0, 0, d])
                else:
                    loc = make_matrix_op([0], [d, 0, 0, d])
        return make_control_op(local_ctrl, loc) if local_ctrl else loc

    def _apply_info(self, info: OpInfo, next_use: Optional[Dict[int, int]] = None,
                    batch: Optional[List[MatrixOp]] = None) -> None:
        """One op.  With `batch`, the localized op is appended instead of applied and the caller flushes the
        list through backend.apply_ops: the runs of local gates between two remaps, so the shard's own
        scheduler (option "tile") sees them together.  The batch is flushed before any remap."""
        self.clock += 1
        if any(self.phys[p] >= self.L for p in info.nondiag_bits):
            if batch:
                self.backend.apply_ops(batch)
                batch.clear()
            self._remap(info.nondiag_bits, next_use)
        for qb in info.ctrl + info.tgt:
            self.last_use[self.n - 1 - qb] = self.clock
        loc = self._localize(info)
        if loc is not None:
            if batch is not None:
                batch.append(loc)
            else:
                self.backend.apply_op(loc)

    def apply_op(self, op: MatrixOp) -> None:
        self._apply_info(OpInfo(self.n, op))

    def apply_ops(self, ops: Iterable[MatrixOp]) -> None:
        self.run_plan(self.plan(list(ops)))

    def plan(self, ops: Sequence[MatrixOp]):
        """Analyse a circuit once: per op the bits that must be local, and for every op index the
        next index at which each logical bit is needed locally (for the farthest-next-use choice)."""
        infos = [OpInfo(self.n, op) for op in ops]
        nxt: List[Dict[int, int]] = [None] * len(infos)
        cur: Dict[int, int] = {}
        for i in range(len(infos) - 1, -1, -1):
            for p in infos[i].nondiag_bits:
                cur[p] = i
            nxt[i] = dict(cur)
        return infos, nxt

    def run_plan(self, plan, batched: bool = False) -> None:
        """batched = False: one launch per gate.  True: the gates between two remaps go to the shard backend in
        one apply_ops call each, so a backend created with tile > 0 applies them as LDS-resident multi-gate
        sweeps (IEEE-equal to gate by gate for tile = 1)."""
        infos, nxt = plan
        batch: Optional[List[MatrixOp]] = [] if batched and hasattr(self.backend, "apply_ops") else None
        for info, nu in zip(infos, nxt):
            self._apply_info(info, nu, batch)
        if batch:
            self.backend.apply_ops(batch)

    # -- reductions ------------------------------------------------------------------------------------------------
    def _allreduce_sum(self, arr: np.ndarray) -> np.ndarray:
        t = self.backend.reduce_tensor(np.ascontiguousarray(arr, dtype=np.float64))
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def _bcast0(self, arr: np.ndarray) -> np.ndarray:
        t = self.backend.reduce_tensor(np.ascontiguousarray(arr, dtype=np.float64))
        self.dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def norm_sqr(self) -> float:
        return float(self._allreduce_sum(np.array([self.backend.norm_sqr()]))[0])

    def measure_probs(self, indices: Sequence[int]) -> np.ndarray:
        """measure_probs (measurement_ops.rs:115-127): bit i of the outcome <-> indices[i]."""
        k = len(indices)
        pps = [self.phys[self.n - 1 - qb] for qb in indices]
        loc = [i for i in range(k) if pps[i] < self.L]
        out = np.zeros(1 << k, dtype=np.float64)
        fixed = 0
        for i in range(k):
            if pps[i] >= self.L:
                fixed |= self._rank_bit(pps[i]) << i
        if loc:
            part = self.backend.measure_probs([self._local_qubit(pps[i]) for i in loc])
            for s, v in enumerate(part):
                m = fixed
                for b, i in enumerate(loc):
                    m |= ((s >> b) & 1) << i
                out[m] += v
        else:
            out[fixed] = self.backend.norm_sqr()
        return self._allreduce_sum(out)

    def measure(self, indices: Sequence[int], measured: Optional[int] = None, rand_u01: float = 0.0):
        """measure (measurement_ops.rs:190-214) on the sharded state.  With `measured` given it plays
        MeasuredCondition; otherwise the outcome is drawn from the marginal distribution of the measured
        qubits with the caller's uniform sample (walking outcomes in increasing order) — statistically the
        reference's soft_measure (:153-176), though not the same sample-to-outcome map, which would need
        the full vector in logical index order.  Collapse: every shard zeroes / rescales with the GLOBAL
        probability (measure_state :220-269; no-op when it is 0)."""
        k = len(indices)
        probs = self.measure_probs(indices)
        if measured is None:
            # every rank must collapse to the SAME outcome: rank 0's sample decides (probs is already identical on
            # all ranks after the all-reduce; the caller's rand_u01 need not be)
            r = float(self._bcast0(np.array([float(rand_u01)]))[0]) * float(probs.sum())
            m = None
            for cand, pm in enumerate(probs):
                if pm > 0:
                    m = cand  # fall-through (rounding left r > 0): the last outcome that can occur, never a p == 0 one
                r -= pm
                if r <= 0 and pm > 0:
                    break
            if m is None:
                m = 0
        else:
            m = int(measured)
        p = float(probs[m])
        if p == 0.0:
            return m, p
        pps = [self.phys[self.n - 1 - qb] for qb in indices]
        loc = [i for i in range(k) if pps[i] < self.L]
        agrees = all(self._rank_bit(pps[i]) == ((m >> i) & 1) for i in range(k) if pps[i] >= self.L)
        if not agrees:
            # this rank's bits contradict the outcome: the whole shard goes to zero
            self.backend.apply_op(make_matrix_op([0], [0, 0, 0, 0]))
        else:
            lm = 0
            for b, i in enumerate(loc):
                lm |= ((m >> i) & 1) << b
            self.backend.measure_state([self._local_qubit(pps[i]) for i in loc], lm, p)
        return m, p

    # -- misc ---------------------------------------------------------------------------------------------------------
    def sync(self) -> None:
        self.backend.sync()

    def set_profile(self, v: int) -> None:
        self.backend.set_profile(v)

    def take_profile(self) -> dict:
        return self.backend.take_profile()

    def comm_stats(self) -> dict:
        """counters since the previous call (they reset), with the summed duration of the collectives"""
        s = dict(self.stats)
        s["collective_ms"] = self.backend.collective_ms()
        for k in self.stats:
            self.stats[k] = 0 if k != "collective_ms" else 0.0
        return s

    def set_tile(self, mode: int) -> None:
        """option "tile" of the shard's state: applies to run_plan(batched=True) batches"""
        shard = getattr(self.backend, "state", None)
        if shard is not None:
            shard.set_option("tile", int(mode))

    def describe(self) -> dict:
        return {"impl": "rustqip_amd.sharded.ShardedState (host planner in Python, torch.distributed all_to_all_single)",
                "world": self.world, "n": self.n, "n_local": self.L}


def make_sharded_state(n: int, n_local: int, dist, device: int, host_staged: bool = False):
    """The sharded state bench.py and the tests drive: the in-library implementation (C ABI qip_hip_dist_*, planner and
    RCCL exchange inside libqip_hip.so) when available, else the torch.distributed one above."""
    return ShardedState(n, dist, backend=HipBackend(n_local, device, host_staged_exchange=host_staged))
