"""Worker for the sharded parity test at BENCH shard size: `world` ranks share ONE MI355X, every rank
holds 2^n_local amplitudes (n_local >= 28: the kernels, grids, tile sweeps and the pack routes run at the shard sizes the
N > 1 bench times), the exchange is the host-staged transport over gloo (RCCL refuses two ranks on one device).  The
comparison itself is oracle/window_parity.sharded_parity — closed sub-cubes of the logical index space gathered through the
layout, a twin sharded state on the literal kernel as the whole-vector guard, closed-form marginals of the product state."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qip_oracle as O  # noqa: E402
from oracle import window_parity as W  # noqa: E402
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402
from rustqip_amd.sharded import DistState  # noqa: E402


def main():
    n_local = int(sys.argv[sys.argv.index("--n-local") + 1]) if "--n-local" in sys.argv else 28
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = world.bit_length() - 1
    n = n_local + g
    # a logical window read back through the layout equals the gathered full vector (small n: both are affordable)
    ns = 14 + g
    xs = circuits.random_state(ns, ns)
    sm = DistState(ns, dist, 0, host_staged=True)
    sm.upload_global(xs)
    sm.apply_ops(circuits.h_layer(ns) + circuits.c2_random_circuit(ns, 40, seed=2) + [q.make_swap_op([0], [ns - 1])])
    full = sm.download_global()
    pick = np.random.default_rng(1).integers(0, 1 << ns, size=5000).astype(np.uint64)
    assert np.array_equal(sm.download_logical(pick), full[pick.astype(np.int64)])
    assert np.array_equal(sm.download(1 << 9, 1 << 10), full[1 << 9:(1 << 9) + (1 << 10)])
    sm.close()

    # (--quick: the gate-by-gate legs at a third of their length — the tile-sweep legs, the remaps and the k_permute_bits pack stay as they are)
    res = W.sharded_parity(lambda: DistState(n, dist, 0, np.complex128, host_staged=True), dist, n, O, q, circuits, gates=256, quick="--quick" in sys.argv)
    if rank == 0:
        brief = {k: v for k, v in res.items() if k != "legs"}
        brief["legs"] = {k: {kk: vv for kk, vv in v.items() if kk in ("gates", "rows", "max_abs_delta", "bit_equal", "ok", "comm", "skipped",
                                                                          "whole_vector_amplitudes_not_equal", "whole_vector_max_abs_delta")}
                         for k, v in res["legs"].items()}
        print("SHARDED_PARITY " + json.dumps(brief))
    assert res["all_legs_ok"], {k: (v["ok"], v["max_abs_delta"], v.get("whole_vector_amplitudes_not_equal")) for k, v in res["legs"].items()}
    assert res["n_local"] == n_local and res["rows_checked"] >= 10**7 and res["gates_skipped"] == 0, res
    assert res["remaps_exercised"] >= 2 and res["packs_via_permute_bits"] >= 1, res
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
