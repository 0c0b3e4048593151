"""Worker for the GPU leg of the N > 1 tests: several ranks share ONE MI355X ("virtual shards").
Shard compute runs the real HIP kernels through HipBackend; the exchange is host-staged over gloo
because RCCL refuses two ranks on one device.  With --nccl and world_size 1 the same script checks
the RCCL process-group plumbing (device tensors, all_reduce) that bench.py --gpus N uses."""
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
from rustqip_amd.sharded import HipBackend, ShardedState  # noqa: E402


def main():
    use_nccl = "--nccl" in sys.argv
    torch.cuda.set_device(0)
    if use_nccl:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = int(math.log2(world))
    for n in (12,):
        x = circuits.random_state(n, n)
        for name, ops in (("c2", circuits.h_layer(n) + circuits.c2_random_circuit(n, 96, seed=28)),
                          ("c4", circuits.c4_clifford_t(n, 96, seed=32)),
                          ("qft", circuits.c3_qft(n)),
                          ("grover_k3", circuits.c5_grover_iteration(n, dense_k3=True))):
            st = ShardedState(n, dist, backend=HipBackend(n - g, 0, host_staged_exchange=not use_nccl))
            st.upload_global(x)
            st.apply_ops(ops)
            got = st.download_global()
            want = O.apply_ops_in_place(n, ops, x.copy())
            err = float(np.max(np.abs(got - want)))
            assert err < 1e-12, (name, n, world, err)
            assert abs(st.norm_sqr() - 1) < 1e-12
            for idx in ([0], [n - 1, 0], list(range(5))):
                assert np.max(np.abs(st.measure_probs(idx) - O.measure_probs(n, idx, want))) < 1e-12
            if world > 1 and name != "grover_k3":
                assert st.stats["remaps"] >= 1
            # the runs of local gates between remaps as LDS-resident tile sweeps on every shard (tile = 1):
            # IEEE-equal to the gate-by-gate shards
            stt = ShardedState(n, dist, backend=HipBackend(n - g, 0, host_staged_exchange=not use_nccl, tile=1))
            stt.upload_global(x)
            stt.run_plan(stt.plan(ops), batched=True)
            assert np.array_equal(stt.download_global(), got), (name, n, world)
            if rank == 0:
                print(f"ok n={n} world={world} {name}: err={err:.2e} stats={st.comm_stats()}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
