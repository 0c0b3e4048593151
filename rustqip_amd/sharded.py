"""State vector sharded over 2^g GPUs, one process per GPU, RCCL over xGMI — a thin caller of the C ABI.

The reference is single-process; its only provision for distribution is the input_offset / output_offset window
arguments (qip-iterators/src/matrix_ops.rs:96-97, qip/src/state_ops/measurement_ops.rs:17-19).  Everything that
decides and moves anything lives in libqip_hip.so (include/qip_hip.h "qip_hip_dist_*", csrc/qip_dist.hip):

  * layout    rank r owns the 2^L (L = n - g) amplitudes whose top g PHYSICAL index bits are r; the planner keeps a
              logical -> physical bit permutation;
  * local     gates whose amplitude-exchanging targets are all physically local run as ordinary local ops; controls and
              diagonal targets on rank bits are resolved per rank (skip / restrict the matrix) and never communicate;
  * exchange  when an exchanging target sits on a rank bit, the g qubits whose next use is farthest are gathered into the
              top g local positions by ONE bit-permutation sweep and ONE all-to-all trades them for the g rank bits
              (ncclGroupStart; ncclSend / ncclRecv x (G-1); ncclGroupEnd inside the library: all xGMI links at once).

This module only (a) binds those entry points (`DistState`), (b) hands the RCCL unique id from rank 0 to the other ranks
over the host's torch.distributed group, and (c) offers a host-staged transport (the C ABI's `qip_hip_transport`
callbacks) so that several ranks can share ONE GPU in tests.  `replay_plan` interprets the library's planner output
(qip_hip_dist_debug_plan) against any shard backend: tests/test_distributed_cpu.py runs it with the CPU oracle and gloo.
"""
from __future__ import annotations

import ctypes as C
import json
import math
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .ops import CircuitError, MatrixOp, make_matrix_op, make_swap_op
from .state import HipState, _check, _u64_array


# ---- a transport staged through host memory (tests: N ranks on one GPU; `dist` is a gloo group) ---------------------
class HostStagedTransport:
    """qip_hip_transport whose all-to-all copies the shard to the host, runs torch.distributed.all_to_all_single there
    (gloo) and copies the result back.  Only for exercising the N > 1 path where RCCL cannot run (two ranks on one
    device); the product transport is the library's built-in RCCL one."""

    def __init__(self, dist, piece_bytes: Optional[int] = None):
        import torch

        self.dist, self.torch = dist, torch
        # None: one all_to_all_single over whole chunks.  A number: the exchange walks the SAME piece list the built-in RCCL
        # transport walks inside its ncclGroupStart / ncclGroupEnd (qip_hip_dist_debug_pieces: every peer in rank order, each
        # chunk in pieces of at most piece_bytes) with one send + one receive per piece, so that loop — which only triggers
        # for >= 2-GiB chunks on real hardware — is exercised at small scale
        self.piece_bytes = piece_bytes
        self.pieces_moved = 0
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self._a2a = _ffi.A2A_FN(self._all_to_all)
        self._ars = _ffi.ARS_FN(self._all_reduce)
        self._a2as = _ffi.A2AS_FN(self._all_to_all_slice)
        self.slices_moved = 0
        self.struct = _ffi.QipTransport(None, self._a2a, self._ars)

    def _all_to_all_slice(self, ctx, send, recv, chunk_bytes, slice_off, slice_bytes, stream):
        """qip_hip_all_to_all_slice_fn (r5): bytes [slice_off, slice_off + slice_bytes) of every chunk.  `stream` is the handle's
        communication stream, which already waits for the sweep part that stores this slice; host-staged, so the host waits too
        (no overlap here — the product transport is RCCL — but the same slices, the same buffers, the same order)."""
        try:
            world = self.dist.get_world_size()
            chunk, off, ln = int(chunk_bytes), int(slice_off), int(slice_bytes)
            h_send = self.torch.empty(ln * world, dtype=self.torch.uint8)
            h_recv = self.torch.empty(ln * world, dtype=self.torch.uint8)
            if self._hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
                return 1
            for c in range(world):
                if self._hip.hipMemcpy(C.c_void_p(h_send.data_ptr() + c * ln), C.c_void_p(send + c * chunk + off), ln, 2) != 0:
                    return 1
            self.dist.all_to_all_single(h_recv, h_send)
            for c in range(world):
                if self._hip.hipMemcpy(C.c_void_p(recv + c * chunk + off), C.c_void_p(h_recv.data_ptr() + c * ln), ln, 1) != 0:
                    return 1
            self.slices_moved += 1
            return 0
        except Exception:  # noqa: BLE001
            return 1

    def _all_to_all(self, ctx, send, recv, chunk_bytes, stream):
        try:
            world = self.dist.get_world_size()
            nbytes = int(chunk_bytes) * world
            h_send = self.torch.empty(nbytes, dtype=self.torch.uint8)
            h_recv = self.torch.empty(nbytes, dtype=self.torch.uint8)
            if self._hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
                return 1
            if self._hip.hipMemcpy(C.c_void_p(h_send.data_ptr()), C.c_void_p(send), nbytes, 2) != 0:  # device -> host
                return 1
            if self.piece_bytes is None:
                self.dist.all_to_all_single(h_recv, h_send)
            else:
                self._exchange_in_pieces(h_send, h_recv, int(chunk_bytes))
            if self._hip.hipMemcpy(C.c_void_p(recv), C.c_void_p(h_recv.data_ptr()), nbytes, 1) != 0:  # host -> device
                return 1
            return 0
        except Exception:  # noqa: BLE001  (nothing may propagate through the C frame)
            return 1

    def _exchange_in_pieces(self, h_send, h_recv, chunk: int) -> None:
        exchange_in_pieces(self.dist, h_send, h_recv, chunk, self.piece_bytes)
        self.pieces_moved += len(piece_list(self.dist.get_rank(), self.dist.get_world_size(), chunk, self.piece_bytes))

    def _all_reduce(self, ctx, values, count):
        try:
            arr = np.ctypeslib.as_array(values, shape=(int(count),))
            t = self.torch.from_numpy(arr.copy())
            self.dist.all_reduce(t)
            arr[:] = t.numpy()
            return 0
        except Exception:  # noqa: BLE001
            return 1


def piece_list(rank: int, world: int, chunk_bytes: int, piece_bytes: int):
    """[(peer, offset, length)]: the library's piece plan for one rank's all-to-all (qip_hip_dist_debug_pieces)"""
    n = int(_ffi.lib.qip_hip_dist_debug_pieces(rank, world, chunk_bytes, piece_bytes, 0, None, None, None))
    if n < 0:
        raise CircuitError(_ffi.last_error())
    peer, off, ln = (C.c_int32 * n)(), (C.c_uint64 * n)(), (C.c_uint64 * n)()
    _ffi.lib.qip_hip_dist_debug_pieces(rank, world, chunk_bytes, piece_bytes, n, peer, off, ln)
    return [(int(peer[i]), int(off[i]), int(ln[i])) for i in range(n)]


def exchange_in_pieces(dist, h_send, h_recv, chunk: int, piece_bytes: int) -> None:
    """all-to-all of `chunk`-byte chunks between host tensors, piece by piece in the library's order: one send + one
    receive per piece, all posted before any is waited for (what the RCCL group does on the device)"""
    rank, world = dist.get_rank(), dist.get_world_size()
    h_recv[rank * chunk:(rank + 1) * chunk] = h_send[rank * chunk:(rank + 1) * chunk]
    ops = []
    for peer, off, ln in piece_list(rank, world, chunk, piece_bytes):
        a = peer * chunk + off
        ops.append(dist.P2POp(dist.isend, h_send[a:a + ln], peer))
        ops.append(dist.P2POp(dist.irecv, h_recv[a:a + ln], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def _broadcast_unique_id(dist, rank: int) -> bytes:
    """rank 0 asks the library for the RCCL unique id; the bytes travel over the host's process group"""
    box = [None]
    if rank == 0:
        buf = (C.c_char * _ffi.QIP_HIP_UNIQUE_ID_BYTES)()
        _check(_ffi.lib.qip_hip_dist_unique_id(buf))
        box[0] = bytes(buf)
    if dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=0)
    return box[0]


class QipLayoutMismatch(RuntimeError):
    """two sharded states that were expected to share a layout do not"""


class DistState:
    """qip_hip_dist: the n-qubit state over dist.get_world_size() ranks (one process per GPU)."""

    def __init__(self, n: int, dist, device: int = 0, dtype=np.complex128, host_staged: bool = False,
                 piece_bytes: Optional[int] = None):
        self.n, self.dist = int(n), dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.g = int(math.log2(self.world))
        if 1 << self.g != self.world:
            raise CircuitError("world size must be a power of two")
        self.L = self.n - self.g
        self.np_dtype = np.dtype(dtype)
        self.dtype = _ffi.QIP_C64 if self.np_dtype == np.complex128 else _ffi.QIP_C32
        self._h = C.c_void_p()
        self._transport = HostStagedTransport(dist, piece_bytes) if host_staged else None
        if host_staged:
            _check(_ffi.lib.qip_hip_dist_create(self.n, self.dtype, device, self.rank, self.world, None,
                                                C.byref(self._transport.struct), C.byref(self._h)))
            _check(_ffi.lib.qip_hip_dist_set_slice_transport(self._h, self._transport._a2as))
        else:
            uid = _broadcast_unique_id(dist, self.rank)
            _check(_ffi.lib.qip_hip_dist_create(self.n, self.dtype, device, self.rank, self.world, uid, None, C.byref(self._h)))
            if piece_bytes is not None:
                _check(_ffi.lib.qip_hip_dist_set_option(self._h, b"piece_bytes", int(piece_bytes)))
        sh = C.c_void_p()
        _check(_ffi.lib.qip_hip_dist_local_state(self._h, C.byref(sh)))
        self.shard = HipState.from_handle(sh, self.L, dtype)  # this rank's 2^L amplitudes (owned by the dist handle)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.lib.qip_hip_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- layout --------------------------------------------------------------------------------------------------
    def layout(self) -> List[int]:
        """phys[p] = physical bit position of logical bit position p (= n-1-qubit)"""
        arr = (C.c_uint32 * self.n)()
        _check(_ffi.lib.qip_hip_dist_layout(self._h, arr))
        return list(arr)

    def rank_flip(self) -> int:
        """pending rank renamings (qip_hip_dist_rank_flip): this rank holds the amplitudes whose rank bits read rank ^ flip"""
        m = C.c_uint32()
        _check(_ffi.lib.qip_hip_dist_rank_flip(self._h, C.byref(m)))
        return int(m.value)

    def logical_indices_of_shard(self) -> np.ndarray:
        """logical index of every local amplitude, in local order"""
        return shard_logical_indices(self.n, self.L, self.rank, self.layout(), self.rank_flip())

    # -- data in / out (tests / small n) -----------------------------------------------------------------------------
    def init_basis(self, logical_index: int) -> None:
        _check(_ffi.lib.qip_hip_dist_init_basis(self._h, int(logical_index)))

    def upload_global(self, x: np.ndarray) -> None:
        """every rank passes the same full logical vector"""
        self.shard.upload(np.ascontiguousarray(x[self.logical_indices_of_shard().astype(np.int64)]))

    def download_global(self) -> np.ndarray:
        """full logical vector on every rank"""
        mine = self.shard.download()
        idx = self.logical_indices_of_shard().astype(np.int64)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, (idx, mine))
        out = np.zeros(1 << self.n, dtype=self.np_dtype)
        for i, v in parts:
            out[i] = v
        return out

    def download_logical(self, indices) -> np.ndarray:
        """amplitudes at the given LOGICAL indices (any order, any subset), the same array on every rank: each rank picks
        what it holds out of its shard with one gather kernel (qip_hip_state_download_indices) and the pieces are put
        together over the host's process group.  Collective.  This is how the parity checks read a window of a sharded
        state at shard sizes where download_global is out of the question."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        owner, local = logical_to_shard(self.n, self.L, self.layout(), self.rank_flip(), idx)
        mine = np.nonzero(owner == np.uint64(self.rank))[0]
        vals = self.shard.download_indices(local[mine]) if mine.size else np.empty(0, dtype=self.np_dtype)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, vals)
        out = np.empty(idx.size, dtype=self.np_dtype)
        for r, v in enumerate(parts):
            out[owner == np.uint64(r)] = v
        return out

    def download(self, offset: int = 0, length: Optional[int] = None) -> np.ndarray:
        """logical amplitudes [offset, offset + length), on every rank (HipState.download's signature: the window checks
        in tests/ and bench.py take either kind of state)"""
        length = (1 << self.n) - offset if length is None else int(length)
        return self.download_logical(np.arange(offset, offset + length, dtype=np.uint64))

    # -- two sharded states side by side (validation support, like HipState.copy_from / max_abs_diff) -----------------
    def _same_layout(self, other: "DistState") -> None:
        if self.layout() != other.layout() or self.rank_flip() != other.rank_flip():
            raise QipLayoutMismatch("the two sharded states are not in the same layout (they must have seen the same batches)")

    def copy_from(self, other: "DistState") -> None:
        """self <- other, shard by shard; the two must be in the same layout (same n, same world, same op batches so far)"""
        self._same_layout(other)
        self.shard.copy_from(other.shard)

    def max_abs_diff(self, other: "DistState"):
        """(max |a_i - b_i|, number of amplitudes that are not IEEE-equal) over the WHOLE sharded vector: every rank
        compares its shard on the device, the figures are combined over the ranks.  Collective."""
        self._same_layout(other)
        worst, differ = self.shard.max_abs_diff(other.shard)
        parts = [None] * self.world
        self.dist.all_gather_object(parts, (worst, differ))
        ws = [w for w, _ in parts]
        return (float("nan") if any(w != w for w in ws) else max(ws)), sum(d for _, d in parts)

    # -- gates -------------------------------------------------------------------------------------------------------------
    def apply_op(self, op: MatrixOp) -> None:
        cop = op.to_c(self.dtype)
        _check(_ffi.lib.qip_hip_dist_apply_op(self._h, C.byref(cop)))

    def compile_ops(self, ops):
        cops = [op.to_c(self.dtype) for op in ops]
        return ((_ffi.QipOp * len(cops))(*cops), cops)

    def apply_compiled(self, compiled) -> None:
        arr, _keep = compiled
        _check(_ffi.lib.qip_hip_dist_apply_ops(self._h, arr, len(arr)))

    def apply_ops(self, ops) -> None:
        self.apply_compiled(self.compile_ops(list(ops)))

    def sync(self) -> None:
        _check(_ffi.lib.qip_hip_dist_sync(self._h))

    def set_option(self, key: str, value: int) -> None:
        _check(_ffi.lib.qip_hip_dist_set_option(self._h, key.encode(), int(value)))

    # -- measurement ------------------------------------------------------------------------------------------------------------
    def norm_sqr(self) -> float:
        out = C.c_double()
        _check(_ffi.lib.qip_hip_dist_norm_sqr(self._h, C.byref(out)))
        return out.value

    def measure_probs(self, indices: Sequence[int]) -> np.ndarray:
        out = np.empty(1 << len(indices), dtype=np.float64)
        _check(_ffi.lib.qip_hip_dist_measure_probs(self._h, _u64_array(indices), len(indices),
                                                   out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def measure(self, indices: Sequence[int], measured: Optional[int] = None, rand_u01: float = 0.0):
        m, p = C.c_uint64(), C.c_double()
        _check(_ffi.lib.qip_hip_dist_measure(self._h, _u64_array(indices), len(indices), -1 if measured is None else int(measured),
                                             float(rand_u01), C.byref(m), C.byref(p)))
        return int(m.value), p.value

    # -- bench / profiling ----------------------------------------------------------------------------------------------------------
    def soft_measure(self, indices: Sequence[int], rand_u01: float) -> int:
        """soft_measure (measurement_ops.rs:153-176) of the sharded state: rank 0's sample decides; no collapse"""
        out = C.c_uint64()
        _check(_ffi.lib.qip_hip_dist_soft_measure(self._h, _u64_array(indices), len(indices), float(rand_u01), C.byref(out)))
        return int(out.value)

    def comm_stats(self) -> dict:
        """counters since the previous call (they reset)"""
        st = _ffi.QipDistStats()
        _check(_ffi.lib.qip_hip_dist_take_stats(self._h, C.byref(st)))
        return {"remaps": int(st.remaps), "pack_sweeps": int(st.pack_sweeps), "bytes_sent_per_rank": int(st.bytes_sent),
                "exchange_ms": st.exchange_ms, "pack_ms": st.pack_ms,
                # read back from the communicator (ncclCommCount / ncclCommUserRank); 0 / -1 with caller-supplied callbacks
                "rccl_ranks": int(st.rccl_ranks), "rccl_rank": int(st.rccl_rank),
                "pieces_sent": int(st.pieces_sent), "piece_bytes": int(st.piece_bytes),
                "packs_via_permute_bits": int(st.packs_via_permute), "packs_folded": int(st.packs_folded),
                # r5, option "dist_overlap": remaps whose exchange ran in slices beside the sweep before them / also the one after
                "remaps_overlapped": int(st.remaps_overlapped), "remaps_overlapped_after": int(st.remaps_overlapped_after),
                "slices_overlapped": int(st.slices_overlapped)}

    def set_profile(self, v: int) -> None:
        self.shard.set_option("profile", int(v))

    def take_profile(self) -> dict:
        prof = self.shard.profile()
        self.shard.profile_reset()
        return prof

    def describe(self) -> dict:
        return {"impl": "libqip_hip.so qip_hip_dist_* (planner + pack sweep + exchange in C++)",
                "transport": "host-staged callbacks (test hook)" if self._transport else "RCCL ncclSend/ncclRecv group (dlopen librccl)",
                "world": self.world, "n": self.n, "n_local": self.L}


def logical_to_shard(n: int, L: int, phys: Sequence[int], flip: int, idx: np.ndarray):
    """(owner rank, index inside that rank's shard) of every logical index — the inverse of shard_logical_indices"""
    idx = np.asarray(idx, dtype=np.uint64)
    P = np.zeros_like(idx)
    for p in range(n):
        P |= ((idx >> np.uint64(p)) & np.uint64(1)) << np.uint64(phys[p])
    owner = (P >> np.uint64(L)) ^ np.uint64(flip)  # rank r holds the physical rank bits r ^ flip (pending renamings)
    return owner, P & np.uint64((1 << L) - 1)


def shard_logical_indices(n: int, L: int, rank: int, phys: Sequence[int], flip: int = 0) -> np.ndarray:
    loc = np.arange(1 << L, dtype=np.uint64)
    P = loc | (np.uint64(rank ^ flip) << np.uint64(L))  # rank bits read as rank ^ flip (pending renamings)
    out = np.zeros_like(P)
    for p in range(n):
        out |= ((P >> np.uint64(phys[p])) & np.uint64(1)) << np.uint64(p)
    return out


# ---- the planner's output, interpreted on the host (test infrastructure for the C++ planner) ----------------------------------
def debug_plan(n: int, rank: int, world: int, ops: Sequence[MatrixOp], dtype: int = _ffi.QIP_C64) -> dict:
    """qip_hip_dist_debug_plan parsed: what `rank` of `world` would do for this circuit on a fresh state (host only)"""
    cops = [op.to_c(dtype) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    txt = _ffi.lib.qip_hip_dist_debug_plan(n, dtype, rank, world, arr, len(cops))
    if not txt:
        raise CircuitError(_ffi.last_error())
    return json.loads(txt.decode() if isinstance(txt, bytes) else txt)


def debug_overlap(n: int, rank: int, world: int, ops: Sequence[MatrixOp], tile_mode: int = 1, slices: int = 4, dtype: int = _ffi.QIP_C64) -> dict:
    """qip_hip_dist_debug_overlap parsed: which remaps of `rank`'s plan the overlapped exchange serves (host only)"""
    cops = [op.to_c(dtype) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    txt = _ffi.lib.qip_hip_dist_debug_overlap(n, dtype, rank, world, arr, len(cops), tile_mode, slices)
    if not txt:
        raise CircuitError(_ffi.last_error())
    return json.loads(txt.decode() if isinstance(txt, bytes) else txt)


def op_from_json(o: dict) -> MatrixOp:
    if o["kind"] == "Matrix":
        return make_matrix_op(o["indices"], [complex(re, im) for re, im in o["data"]])
    if o["kind"] == "SparseMatrix":
        return MatrixOp.new_sparse(o["indices"], [[(int(c), complex(re, im)) for c, re, im in row] for row in o["rows"]])
    if o["kind"] == "Swap":
        h = o["half"]
        return make_swap_op(o["indices"][:h], o["indices"][h:])
    inner = op_from_json(o["inner"])
    return MatrixOp.new_control(o["indices"][: o["n_controls"]], o["indices"][o["n_controls"]:], inner)


def pack_bits_numpy(x: np.ndarray, L: int, sel: Sequence[int]) -> np.ndarray:
    """the pack sweep on the host: destination bit L-g+t <- source bit sel[t]; every other bit keeps its relative order"""
    g = len(sel)
    j = np.arange(1 << L, dtype=np.uint64)
    rest, c = j & np.uint64((1 << (L - g)) - 1), j >> np.uint64(L - g)
    src = np.zeros_like(j)
    keep = [p for p in range(L) if p not in sel]
    for i, p in enumerate(keep):
        src |= ((rest >> np.uint64(i)) & np.uint64(1)) << np.uint64(p)
    for t, p in enumerate(sel):
        src |= ((c >> np.uint64(t)) & np.uint64(1)) << np.uint64(p)
    return x[src.astype(np.int64)]


def replay_plan(plan: dict, shard: np.ndarray, apply_local, all_to_all) -> np.ndarray:
    """Run a debug plan: `apply_local(L, op, shard) -> shard`, `all_to_all(send) -> recv` (equal chunks in rank order)."""
    L = plan["L"]
    for st in plan["steps"]:
        if st["t"] == "local":
            shard = apply_local(L, op_from_json(st["op"]), shard)
        elif st["t"] == "pack":
            shard = pack_bits_numpy(shard, L, st["sel"])
        else:
            shard = all_to_all(shard)
    return shard
