"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with gloo.

The C++ planner of the sharded state (csrc/qip_dist.hip: logical -> physical map, farthest-next-use remap choice,
per-rank localisation of every op) is pure host code; qip_hip_dist_debug_plan serialises what THIS rank would do.
Here the plan is replayed with the CPU oracle as the shard and gloo as the transport, and the gathered result is
compared with the oracle applied to the full vector — the N > 1 logic without any GPU."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qip_oracle as O  # noqa: E402
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402
from rustqip_amd import sharded  # noqa: E402


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def mixed_ops(n, rng, count):
    ops = []
    for _ in range(count):
        kind = int(rng.integers(0, 9))
        perm = [int(v) for v in rng.permutation(n)]
        if kind == 0:
            ops.append(q.make_matrix_op([perm[0]], circuits.H))
        elif kind == 1:
            ops.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], circuits.X)))
        elif kind == 2:
            ops.append(q.make_control_op([perm[0]], q.make_matrix_op([perm[1]], [1, 0, 0, np.exp(0.3j)])))
        elif kind == 3:
            ops.append(q.make_swap_op([perm[0]], [perm[1]]))
        elif kind == 4:
            ops.append(q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel()))
        elif kind == 5:
            d = np.exp(1j * rng.uniform(0, 6, 8))
            ops.append(q.make_matrix_op(perm[:3], np.diag(d).ravel()))
        elif kind == 6:
            rows = [[(1, 0.5j)], [(0, 2.0)], [(3, 1.0)], [(2, -1.0), (3, 0.25)]]
            ops.append(q.make_sparse_matrix_op(perm[:2], rows))
        elif kind == 7:
            ops.append(q.make_control_op([perm[0]], q.make_swap_op([perm[1]], [perm[2]])))
        else:  # block-diagonal in its first target, dense in the second: the first may stay on a rank bit
            u0, u1 = rand_unitary(1, rng), rand_unitary(1, rng)
            m = np.zeros((4, 4), dtype=complex)
            m[:2, :2], m[2:, 2:] = u0, u1
            ops.append(q.make_matrix_op(perm[:2], m.ravel()))
    return ops


def xy_ops(n, rng, count):
    """X / Y walls and singles on every qubit (on a rank bit they only rename the ranks), interleaved with gates that read
    those qubits as controls, as diagonal targets and as dense targets (the renaming has to be honoured / settled)"""
    Y = [0, -1j, 1j, 0]
    A = [0, 0.6 + 0.8j, 1j, 0]  # a general anti-diagonal gate
    ops = []
    for _ in range(count):
        kind = int(rng.integers(0, 10))
        perm = [int(v) for v in rng.permutation(n)]
        if kind == 8:  # an uncontrolled swap only renames qubits — also while a rank renaming is pending on one of them
            ops.append(q.make_swap_op([perm[0]], [perm[1]]))
        elif kind == 9:
            ops.append(q.make_swap_op(perm[:2], perm[2:4]))
        elif kind <= 2:
            ops.append(q.make_matrix_op([perm[0]], [circuits.X, Y, A][kind]))
        elif kind == 3:
            ops.append(q.make_control_op([perm[0]], q.make_matrix_op([perm[1]], circuits.X)))
        elif kind == 4:
            ops.append(q.make_control_op(perm[:2], q.make_matrix_op([perm[2]], [1, 0, 0, np.exp(0.7j)])))
        elif kind == 5:
            ops.append(q.make_matrix_op([perm[0]], circuits.rz(0.3 + 0.1 * perm[0])))
        elif kind == 6:
            ops.append(q.make_matrix_op([perm[0]], circuits.H))
        else:
            ops += [q.make_matrix_op([t], circuits.X) for t in range(n)]  # a wall
    return ops


def apply_local(L, op, shard):
    out = np.zeros_like(shard)
    O.apply_op_overwrite(L, op, np.ascontiguousarray(shard), out)
    return out


def all_to_all(send):
    t_send = torch.from_numpy(np.ascontiguousarray(send).view(np.float64))
    t_recv = torch.empty_like(t_send)
    dist.all_to_all_single(t_recv, t_send)
    return t_recv.numpy().view(np.complex128)


def all_to_all_in_pieces(send):
    """the same exchange through the library's piece list (qip_hip_dist_debug_pieces): what the built-in RCCL transport
    walks inside its send / receive group, here with ragged pieces of about a third of a chunk over gloo"""
    world = dist.get_world_size()
    t_send = torch.from_numpy(np.ascontiguousarray(send).view(np.uint8))
    t_recv = torch.empty_like(t_send)
    chunk = t_send.numel() // world
    piece = max(16, (chunk // 3) // 16 * 16)
    sharded.exchange_in_pieces(dist, t_send, t_recv, chunk, piece)
    return t_recv.numpy().view(np.complex128)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = int(math.log2(world))
    # (an exchange moves all g rank bits at once, so an op with k non-diagonal targets needs L >= g + k local qubits:
    # three rank bits want shards of at least 2^6)
    sizes = ((g + max(g, 3), 0), (7, 1), (9, 2)) if g <= 2 else ((2 * g + 3, 0), (2 * g + 4, 1), (2 * g + 5, 2))
    for n, seed in sizes:
        rng = np.random.default_rng(seed)  # same stream on every rank
        x = circuits.random_state(n, seed + 10)
        L = n - g
        for name, ops in (("c4", circuits.h_layer(n) + circuits.c4_clifford_t(n, 64, seed=32)),
                          ("qft", circuits.c3_qft(n)),
                          ("grover", circuits.c5_grover_iteration(n)),
                          ("grover_k3", circuits.c5_grover_iteration(n, dense_k3=True)),
                          ("mixed", mixed_ops(n, rng, 60)),
                          ("xy", xy_ops(n, rng, 80))):
            plan = sharded.debug_plan(n, rank, world, ops)
            assert (plan["g"], plan["L"], plan["rank"]) == (g, L, rank)
            shard = x[rank << L:(rank + 1) << L].copy()  # a fresh state: logical = physical
            shard = sharded.replay_plan(plan, shard, apply_local, all_to_all_in_pieces if name in ("qft", "mixed", "grover_k3") else all_to_all)
            idx = sharded.shard_logical_indices(n, L, rank, plan["phys"], plan["flip"]).astype(np.int64)
            parts = [None] * world
            dist.all_gather_object(parts, (idx, shard, [(s["t"], s.get("sel")) for s in plan["steps"] if s["t"] != "local"], plan["phys"]))
            got = np.zeros(1 << n, dtype=np.complex128)
            for i, v, _, _ in parts:
                got[i] = v
            want = O.apply_ops_in_place(n, ops, x.copy())
            err = float(np.max(np.abs(got - want)))
            assert err < 1e-12, (name, n, world, err)
            # SPMD: every rank packs / exchanges at the same places with the same selection, and ends in the same layout
            for _, _, coll, phys in parts:
                assert coll == parts[0][2] and phys == parts[0][3], (name, n)
            flips = [None] * world
            dist.all_gather_object(flips, plan["flip"])
            assert all(f == flips[0] for f in flips)  # the renaming of the ranks is the same decision everywhere
            assert sorted(plan["phys"]) == list(range(n))
            n_exchange = sum(1 for s in plan["steps"] if s["t"] == "exchange")
            n_pack = sum(1 for s in plan["steps"] if s["t"] == "pack")
            if name in ("c4", "qft"):
                assert n_exchange >= 1, "circuit was expected to touch a global qubit"
            assert n_pack <= n_exchange
            # every local op only names local qubits
            for s in plan["steps"]:
                if s["t"] == "local":
                    assert all(0 <= i < L for i in s["op"]["indices"]), s
            if rank == 0:
                print(f"ok n={n} world={world} {name}: err={err:.2e} exchanges={n_exchange} packs={n_pack}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
