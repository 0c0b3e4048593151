"""N > 1 path on CPU: world_size 2, 4 and 8 over gloo.  The sharded state's planner lives in libqip_hip.so
(qip_hip_dist_debug_plan); the worker replays its plans with the CPU oracle as the shard (tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd import circuits, sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_state_matches_single_process(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("ok n=") == 18


def test_planner_at_bench_size_without_any_device():
    """the plans of the BASELINE multi-GPU configs at their stated sizes (host arithmetic only): n = 33 over 8 ranks,
    n = 32 over 4: few exchanges, at most one pack sweep each, every rank the same collective sequence"""
    for n, world, ops in ((33, 8, circuits.c2_random_circuit(33, 256, seed=28, single_only=True)),
                          (32, 4, circuits.c4_clifford_t(32, 256, seed=32)),
                          (33, 8, circuits.c5_grover_iteration(33, dense_k3=True))):
        seqs = []
        for rank in range(world):
            plan = sharded.debug_plan(n, rank, world, ops)
            seqs.append([(s["t"], s.get("sel")) for s in plan["steps"] if s["t"] != "local"])
        assert all(sq == seqs[0] for sq in seqs)
        ex = sum(1 for t, _ in seqs[0] if t == "exchange")
        assert 1 <= ex <= 12, (n, world, ex)
        assert sum(1 for t, _ in seqs[0] if t == "pack") <= ex


def test_planner_chooses_the_leaving_qubits_by_modelled_cost():
    """r4: when the count-optimal leaving set (farthest next use) would gather from a position inside a wave row — a gather
    that can never ride in the preceding tile sweep's store — the planner also rolls the rest of the circuit forward with the row
    positions excluded and keeps the cheaper modelled total (exchange = shard / world bytes per link at 153 GB/s, free-standing
    gather = one copy of the shard).  Never more exchanges; fewer gathers from row positions; option 0 = the old rule."""
    rows = {}
    for name, n, world, ops in (("grover", 33, 8, circuits.c5_grover_iteration(33)),
                                ("clifford_t", 32, 4, circuits.c4_clifford_t(32, 256)),
                                ("mixed_x4", 32, 4, circuits.c2_random_circuit(32, 1024, seed=5)),
                                ("qft", 33, 8, circuits.c3_qft(33))):
        try:
            q.set_global_option("dist_plan_cost", 0)
            old = sharded.debug_plan(n, 0, world, ops)["model"]
        finally:
            q.set_global_option("dist_plan_cost", 1)
        plans = [sharded.debug_plan(n, r, world, ops) for r in (0, world - 1)]
        new = plans[0]["model"]
        assert [s for s in plans[0]["steps"] if s["t"] != "local"] == [s for s in plans[1]["steps"] if s["t"] != "local"]  # same collectives on every rank
        cost = lambda m: m["exchanges"] * m["exchange_ms"] + m["packs_from_row_positions"] * m["pack_ms"]
        assert new["exchanges"] <= old["exchanges"] and cost(new) <= cost(old), (name, old, new)
        rows[name] = (old["packs_from_row_positions"], new["packs_from_row_positions"])
        g, L = plans[0]["g"], plans[0]["L"]
        assert new["row_p5"] == 11 and abs(new["exchange_ms"] - 1e3 * 16 * 2**L / world / 153e9) < 1e-3
    assert rows["grover"] == (2, 0) and rows["clifford_t"][1] < rows["clifford_t"][0] and rows["mixed_x4"][1] < rows["mixed_x4"][0], rows


def test_pack_bits_model_and_invalid_worlds():
    x = np.arange(64, dtype=np.complex128)
    y = sharded.pack_bits_numpy(x, 6, [1, 4])  # bits 1 and 4 become bits 4 and 5
    for j in range(64):
        src = ((j >> 4) & 1) << 1 | ((j >> 5) & 1) << 4
        rest = j & 15
        keep = [0, 2, 3, 5]
        for i, p in enumerate(keep):
            src |= ((rest >> i) & 1) << p
        assert y[j] == x[src]
    with pytest.raises(q.CircuitError):
        sharded.debug_plan(6, 0, 3, [q.make_matrix_op([0], circuits.H)])
    with pytest.raises(q.CircuitError):
        sharded.debug_plan(3, 0, 4, [q.make_matrix_op([0], circuits.H)])


def test_piece_plan_of_the_exchange():
    """qip_hip_dist_debug_pieces — the list the built-in RCCL transport walks inside one send / receive group: every peer,
    every byte of every chunk exactly once, no piece above the limit, pieces of one peer in offset order, and the list of
    rank r towards peer p mirrors the list of p towards r (sends meet their receives in order).  At bench size: a 16-GiB
    shard over 8 ranks has 2-GiB chunks = two 1-GiB pieces per peer."""
    for world, chunk, piece in ((2, 96, 32), (4, 1000 * 16, 48 * 16), (8, 1 << 31, 1 << 30), (8, 5 * 16, 1 << 30), (4, 64, 16)):
        lists = [sharded.piece_list(r, world, chunk, piece) for r in range(world)]
        for r, ps in enumerate(lists):
            assert all(p != r and 0 < ln <= piece for p, _, ln in ps)
            for p in range(world):
                mine = [(off, ln) for pp, off, ln in ps if pp == p]
                if p == r:
                    assert not mine
                    continue
                assert [off for off, _ in mine] == sorted(off for off, _ in mine)
                assert sum(ln for _, ln in mine) == chunk and mine[0][0] == 0
                assert all(a[0] + a[1] == b[0] for a, b in zip(mine, mine[1:]))
                assert mine == [(off, ln) for pp, off, ln in lists[p] if pp == r]
    assert len(sharded.piece_list(3, 8, 1 << 31, 1 << 30)) == 14
    assert sharded.piece_list(0, 1, 1 << 20, 1 << 10) == []


def test_uncontrolled_swap_is_a_relabelling():
    """Swap(h, A ++ B) without controls moves no amplitude on a sharded state: the planner exchanges the qubits' entries of
    the logical -> physical map, rank bits included (SwapOpIterator, qubit_iterators.rs:176-219, is a pure index-bit
    permutation); a controlled swap is still executed"""
    n, world = 10, 4
    ops = [q.make_swap_op([0], [9]), q.make_swap_op([1, 2], [7, 3])]
    for rank in range(world):
        plan = sharded.debug_plan(n, rank, world, ops)
        assert plan["steps"] == []
        want = list(range(n))  # phys[p] for logical bit p = n - 1 - qubit
        for a, b in ((0, 9), (1, 7), (2, 3)):
            want[n - 1 - a], want[n - 1 - b] = want[n - 1 - b], want[n - 1 - a]
        assert plan["phys"] == want
    plan = sharded.debug_plan(n, 1, world, [q.make_control_op([5], q.make_swap_op([0], [9]))])
    assert [s["t"] for s in plan["steps"]].count("exchange") == 1
    # the closing bit reversal of a QFT costs nothing: same exchanges with and without it
    qft = circuits.c3_qft(12)
    with_swaps = sharded.debug_plan(12, 0, 4, qft)
    without = sharded.debug_plan(12, 0, 4, [o for o in qft if o.kind != "Swap"])
    count = lambda p, t: sum(1 for s in p["steps"] if s["t"] == t)
    assert count(with_swaps, "exchange") == count(without, "exchange") and count(with_swaps, "local") == count(without, "local")


def test_logical_windows_of_a_sharded_state_index_math():
    """DistState.download_logical reads a LOGICAL window out of the ranks' shards (the sharded parity checks at bench shard
    size): its index map must be the inverse of shard_logical_indices for every layout and pending rank renaming."""
    from rustqip_amd.sharded import logical_to_shard, shard_logical_indices

    rng = np.random.default_rng(3)
    for n, g in ((9, 1), (10, 2), (11, 3)):
        L = n - g
        for _ in range(5):
            phys = [int(v) for v in rng.permutation(n)]
            flip = int(rng.integers(0, 1 << g))
            seen = np.zeros(1 << n, dtype=np.int64)
            for rank in range(1 << g):
                logical = shard_logical_indices(n, L, rank, phys, flip)
                owner, local = logical_to_shard(n, L, phys, flip, logical)
                assert np.all(owner == rank) and np.array_equal(local, np.arange(1 << L, dtype=np.uint64))
                seen[logical.astype(np.int64)] += 1
            assert np.all(seen == 1)  # the shards tile the logical index space exactly once


def test_overlapped_exchange_is_cut_where_every_rank_can_cut_it():
    """r5, option dist_overlap: the positions that cut a remap's exchange into slices must be the same on every rank although the ranks'
    local batches — and so the tiles of their edge sweeps — differ.  The host-only predicate (qip_hip_dist_debug_overlap: every rank's plan,
    votes summed as the executor's all-reduce sums them) answers alike whichever rank asks, picks positions below the chunk-selecting ones
    and above the rows, and `after` implies `before`.  Bench circuits at bench size (n = 32 / 33 over 4 / 8 ranks): the model's input."""
    import math

    from rustqip_amd import circuits, sharded

    seen_before = seen_after = 0
    for name, gen in (("c2", lambda n: circuits.c2_random_circuit(n, 256, seed=28)), ("c4", lambda n: circuits.c4_clifford_t(n, 256, seed=32)),
                      ("grover", lambda n: circuits.c5_grover_iteration(n)), ("qft", lambda n: circuits.c3_qft(n))):
        for world, slices, mode in ((4, 4, 1), (8, 4, 1 | 16), (2, 2, 2), (8, 8, 1)):
            g = int(math.log2(world))
            n = 30 + g
            ops = gen(n)
            plans = [sharded.debug_overlap(n, r, world, ops, mode, slices)["remaps"] for r in ((0, world - 1) if world > 2 else (0, 1))]
            exchanges = sharded.debug_plan(n, 0, world, ops)["model"]["exchanges"]
            assert all(len(p) == exchanges for p in plans), (name, world)
            for a, b in zip(plans[0], plans[1]):
                assert (a["pack"], a["before"], a["positions"]) == (b["pack"], b["before"], b["positions"]), (name, world, a, b)
                if a["before"]:
                    assert len(a["positions"]) == int(math.log2(slices)) and a["positions"] == sorted(set(a["positions"]))
                    assert all(12 <= q < n - 2 * g for q in a["positions"]), a
                else:
                    assert not a["positions"] and not a["after"]
                for x in (a, b):
                    assert not x["after"] or x["before"]
                seen_before += a["before"]
                seen_after += a["after"]
    assert seen_before >= 12 and seen_after >= 8, (seen_before, seen_after)
