#!/usr/bin/env python
"""The MODELLED weak-scaling table of DESIGN.md §5 (no multi-GPU hardware has been available in any round: this is host arithmetic
over the planner's own plans, not a measurement).

  python tools/model_scaling.py [n_local]        (default 30: the bench's shard size)

Per circuit and world size N: the planner's plan for rank 0 at n = n_local + log2 N (qip_hip_dist_debug_plan: exchanges, gathers,
which of them select a position inside a wave row), priced with the plan's own model (exchange = shard / N bytes over each of the N - 1
xGMI links at 153 GB/s; a gather that cannot ride in a tile sweep = one copy of the shard at 6.2 TB/s) and the MEASURED single-GPU time
per gate at this shard size (profiles/r04_bench_n1.json: gate by gate 5.27 ms; as tile sweeps from the bench line's medians)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits, sharded  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    nl = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    line = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")).read().strip().splitlines()[-1])
    ex = line["extras"]
    t_gate = line["ms_per_step"] / line["config"]["gates_per_step"]  # ms per gate, gate by gate, n = 30
    # measured single-GPU medians (ms) for the whole circuit: gate by gate / best IEEE-equal sweeps / best 1e-12 sweeps
    single = {
        "headline (256 H / X / Rz)": (line["ms_per_step"], None, None),
        "configs[1] mix (256)": (line["mixed_circuit"]["ms"], ex["tiled_mode1_jit_wide_relabel"]["ms"], ex["tiled_mode2_jit_fma_merge_wide"]["ms"]),
        "Clifford+T (256)": (ex["configs3_clifford_t_n30"]["ms"], ex["configs3_clifford_t_n30"]["tile1_jit_wide_relabel"]["ms"],
                             ex["configs3_clifford_t_n30"]["tile2_jit_fma_merge_wide_relabel"]["ms"]),
        "Grover iteration (182)": (ex["configs4_grover_iteration_n30"]["ms"], ex["configs4_grover_iteration_n30"]["tile1_jit_wide"]["ms"], None),
        "QFT (480)": (ex["configs2_qft_n30"]["ms"], ex["configs2_qft_n30"]["tile1_jit"]["ms"], ex["configs2_qft_n30"]["tile2_jit_fma_merge_wide"]["ms"]),
    }
    print(f"MODEL, not a measurement.  n_local = {nl}; single-GPU times measured at n = 30 (profiles/r04_bench_n1.json, {t_gate:.2f} ms per gate gate by gate).\n")
    print("| circuit | N | n | exchanges | gathers (from a row position) | modelled comm ms | gate by gate: ms, per-GPU efficiency | IEEE-equal sweeps | 1e-12 sweeps |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, gen in (("headline (256 H / X / Rz)", lambda n: circuits.c2_random_circuit(n, 256, seed=28, single_only=True)),
                      ("configs[1] mix (256)", lambda n: circuits.c2_random_circuit(n, 256, seed=28)),
                      ("Clifford+T (256)", lambda n: circuits.c4_clifford_t(n, 256, seed=32)),
                      ("Grover iteration (182)", lambda n: circuits.c5_grover_iteration(n)),
                      ("QFT (480)", lambda n: circuits.c3_qft(n))):
        for world in (1, 2, 4, 8):
            g = world.bit_length() - 1
            n = nl + g
            ops = gen(n)
            if world == 1:
                exch = packs = rows = 0
                comm_lo = comm_hi = 0.0
            else:
                m = sharded.debug_plan(n, 0, world, ops)["model"]
                exch, packs, rows = m["exchanges"], m["packs"], m["packs_from_row_positions"]
                comm_lo = exch * m["exchange_ms"] + rows * m["pack_ms"]   # every other gather rides in a tile sweep
                comm_hi = exch * m["exchange_ms"] + packs * m["pack_ms"]  # gate by gate: every gather is a sweep of its own
            # local work per rank does not grow with N in this weak scaling (a gate sweeps the 2^n_local shard; gates whose
            # exchanging target sits on a rank bit are served after a remap)
            cols = []
            for idx, comm in ((0, comm_hi), (1, comm_lo), (2, comm_lo)):
                t1 = single[name][idx]
                if t1 is None:
                    cols.append("—")
                else:
                    # n_local = 30 scaling of the measured n = 30 time: proportional to the shard size
                    t1s = t1 * (2.0 ** (nl - 30))
                    cols.append(f"{t1s + comm:.0f} ms, {100 * t1s / (t1s + comm):.0f} %")
            print(f"| {name} | {world} | {n} | {exch} | {packs} ({rows}) | {comm_lo:.0f} – {comm_hi:.0f} | " + " | ".join(cols) + " |")


if __name__ == "__main__":
    main()
