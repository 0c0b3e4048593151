"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with gloo."""
import math
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cpu_backend import OracleBackend  # noqa: E402
from oracle import qip_oracle as O  # noqa: E402
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402
from rustqip_amd.sharded import ShardedState  # noqa: E402


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def mixed_ops(n, rng, count):
    ops = []
    for _ in range(count):
        kind = int(rng.integers(0, 8))
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
        else:
            ops.append(q.make_control_op([perm[0]], q.make_swap_op([perm[1]], [perm[2]])))
    return ops


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = int(math.log2(world))
    for n, seed in ((g + 3, 0), (7, 1), (9, 2)):
        rng = np.random.default_rng(seed)  # same stream on every rank
        x = circuits.random_state(n, seed + 10)
        for name, ops in (("c4", circuits.h_layer(n) + circuits.c4_clifford_t(n, 64, seed=32)),
                          ("qft", circuits.c3_qft(n)),
                          ("grover", circuits.c5_grover_iteration(n)),
                          ("mixed", mixed_ops(n, rng, 60))):
            st = ShardedState(n, dist, backend=OracleBackend(n - g))
            st.upload_global(x)
            if name == "mixed":
                for op in ops:  # un-planned path (least-recently-used choice)
                    st.apply_op(op)
            else:
                st.apply_ops(ops)  # planned path (farthest-next-use choice)
            got = st.download_global()
            want = O.apply_ops_in_place(n, ops, x.copy())
            err = float(np.max(np.abs(got - want)))
            assert err < 1e-12, (name, n, world, err)
            assert abs(st.norm_sqr() - O.prob_magnitude(want)) < 1e-12
            for idx in ([0], [n - 1], [0, n - 1, 2], list(range(n))):
                assert np.max(np.abs(st.measure_probs(idx) - O.measure_probs(n, idx, want))) < 1e-12, (name, idx)
            if name in ("c4", "qft"):
                assert st.stats["remaps"] >= 1, "circuit was expected to touch a global qubit"
                # batched replay: the runs of local gates between remaps reach the backend as lists (what lets a
                # HIP shard apply them as tile sweeps); same ops in the same order, so the result is identical
                stb = ShardedState(n, dist, backend=OracleBackend(n - g))
                stb.upload_global(x)
                stb.run_plan(stb.plan(ops), batched=True)
                assert np.array_equal(stb.download_global(), got), (name, n)
                assert 1 <= len(stb.backend.batches) <= stb.stats["remaps"] + 1
                assert (stb.stats["remaps"], stb.stats["local_swaps"]) == (st.stats["remaps"], st.stats["local_swaps"])
            # collapsing measurement, forced outcomes: every shard rescales with the global probability
            for idx, forced in (([0], 1), ([n - 1, 1], 2), ([2, 0, n - 1], 5)):
                st2 = ShardedState(n, dist, backend=OracleBackend(n - g))
                st2.upload_global(x)
                st2.apply_ops(ops[: len(ops) // 2])
                ref = O.apply_ops_in_place(n, ops[: len(ops) // 2], x.copy())
                m, p = st2.measure(idx, measured=forced)
                out = np.zeros_like(ref)
                wm, wp = O.measure(n, idx, ref, out, forced=forced)
                if wp == 0:
                    out = ref
                assert m == wm and abs(p - wp) < 1e-12, (name, idx)
                assert np.max(np.abs(st2.download_global() - out)) < 1e-12, (name, idx)
                ms, ps = st2.measure(idx, rand_u01=0.37)  # sampled: already collapsed, so it repeats
                if wp > 0:
                    assert ms == forced and abs(ps - 1) < 1e-12
            # a basis state finds its owner through the permuted layout
            st.init_basis(5 % (1 << n))
            e = np.zeros(1 << n, dtype=np.complex128)
            e[5 % (1 << n)] = 1
            assert np.array_equal(st.download_global(), e)
            if rank == 0:
                print(f"ok n={n} world={world} {name}: err={err:.2e} stats={st.stats}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
