#!/usr/bin/env python
"""The MODELLED weak-scaling table of DESIGN.md §5 (no multi-GPU hardware has been available in any round: this is host arithmetic
over the planner's own plans, not a measurement).

  python tools/model_scaling.py [bench line file = newest profiles/r0*_bench_n1.json] [n_local = 30]

Per circuit and world size N: the planner's plan for rank 0 at n = n_local + log2 N (qip_hip_dist_debug_plan: exchanges, gathers, which of
them select a position inside a wave row), priced with the plan's own model (exchange = shard / N bytes over each of the N - 1 xGMI links at
153 GB/s; a gather that cannot ride in a tile sweep = one copy of the shard at 6.2 TB/s) and the MEASURED single-GPU times of the bench line.

r5: the overlapped exchange (option "dist_overlap", P = 4 slices).  Which remaps it serves comes from the library's own predicate
(qip_hip_dist_debug_overlap: the edge sweeps of the two batches scheduled as apply_ops schedules them); a served remap is priced as a
pipeline of P slices through (sweep part, exchange slice[, sweep part]):  T = a + b [+ a'] + (P - 1) * max(a, b[, a']),  a = C * S / P,
b = E / P, with S the mode's measured time per sweep, E the modelled exchange and C = 1.25 for the HBM bandwidth the exchange takes from a
sweep that runs beside it (N = 8: 2 x 7 x 153 GB/s = 2.1 of ~6.5 TB/s).  What the overlap hides is (S + E [+ S']) - T."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402,F401
from rustqip_amd import circuits, sharded  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, CONTENTION = 4, 1.25


def hidden(E, S, before, after):
    """milliseconds of (sweep + exchange [+ sweep]) that the P-slice pipeline hides"""
    if not before:
        return 0.0
    a, b = CONTENTION * S / P, E / P
    if after:
        return max(0.0, (S + E + S) - (a + b + a + (P - 1) * max(a, b)))
    return max(0.0, (S + E) - (a + b + (P - 1) * max(a, b)))


def main():
    args = [a for a in sys.argv[1:]]
    path = args[0] if args and os.path.exists(args[0]) else sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_n1.json")))[-1]
    nl = int(args[-1]) if args and args[-1].isdigit() else 30
    line = json.loads(open(path).read().strip().splitlines()[-1])
    ex = line["extras"]
    t_gate = line["ms_per_step"] / line["config"]["gates_per_step"]

    def leg(d):  # (ms of the whole circuit, ms per sweep)
        return (d["ms"], d["ms"] / max(1, d["launches"])) if d else None

    c3, c4g, qft = ex["configs3_clifford_t_n30"], ex["configs4_grover_iteration_n30"], ex["configs2_qft_n30"]
    # mode -> (scheduler mode bits for the predicate, {circuit: (ms, ms per sweep)})
    modes = {
        "interpreter sweeps (tile = 1)": (1, {"configs[1] mix (256)": leg(ex["tiled_mode1"]), "Clifford+T (256)": leg(c3["tile1"]),
                                              "Grover iteration (182)": leg(c4g["tile1"]), "QFT (480)": leg(qft["tile1"])}),
        "compiled wide sweeps (tile = 1, IEEE-equal)": (1 | 16, {"configs[1] mix (256)": leg(ex["tiled_mode1_jit_wide"]), "Clifford+T (256)": leg(c3.get("tile1_jit_wide")),
                                                                   "Grover iteration (182)": leg(c4g.get("tile1_jit_wide")), "QFT (480)": leg(qft["tile1_jit"])}),
        "compiled wide sweeps, 1e-12 mode (tile = 2, fma, merged runs)": (2 | 16, {"configs[1] mix (256)": leg(ex["tiled_mode2_jit_fma_merge_wide"]),
                                                                                     "Clifford+T (256)": leg(c3.get("tile2_jit_fma_merge_wide_relabel")),
                                                                                     "QFT (480)": leg(qft.get("tile2_jit_fma_merge_wide"))}),
    }
    gens = (("headline (256 H / X / Rz)", lambda n: circuits.c2_random_circuit(n, 256, seed=28, single_only=True)),
            ("configs[1] mix (256)", lambda n: circuits.c2_random_circuit(n, 256, seed=28)),
            ("Clifford+T (256)", lambda n: circuits.c4_clifford_t(n, 256, seed=32)),
            ("Grover iteration (182)", lambda n: circuits.c5_grover_iteration(n)),
            ("QFT (480)", lambda n: circuits.c3_qft(n)))
    gbg = {"headline (256 H / X / Rz)": line["ms_per_step"], "configs[1] mix (256)": line["mixed_circuit"]["ms"], "Clifford+T (256)": c3["ms"],
           "Grover iteration (182)": c4g["ms"], "QFT (480)": qft["ms"]}
    print(f"MODEL, not a measurement.  n_local = {nl}; single-GPU times measured at n = 30 ({os.path.relpath(path, ROOT)}, {t_gate:.2f} ms per gate gate by gate).  "
          f"Overlap: P = {P} slices, contention factor {CONTENTION}.\n")
    print("### gate by gate (the BASELINE metric)\n")
    print("| circuit | N | n | exchanges | gathers (from a row position) | modelled comm ms | ms, per-GPU efficiency |")
    print("|---|---|---|---|---|---|---|")
    plans = {}
    for name, gen in gens:
        for world in (1, 2, 4, 8):
            g = world.bit_length() - 1
            n = nl + g
            if world == 1:
                m = {"exchanges": 0, "packs": 0, "packs_from_row_positions": 0, "exchange_ms": 0.0, "pack_ms": 0.0}
            else:
                m = sharded.debug_plan(n, 0, world, gen(n))["model"]
            plans[(name, world)] = m
            comm = m["exchanges"] * m["exchange_ms"] + m["packs"] * m["pack_ms"]  # gate by gate: every gather is a sweep of its own
            t1 = gbg[name] * 2.0 ** (nl - 30)
            print(f"| {name} | {world} | {n} | {m['exchanges']} | {m['packs']} ({m['packs_from_row_positions']}) | {comm:.0f} | {t1 + comm:.0f} ms, {100 * t1 / (t1 + comm):.0f} % |")
    for mode_name, (bits, single) in modes.items():
        print(f"\n### {mode_name}\n")
        print("| circuit | N | remaps: served before / also after | comm ms serial (gathers ride in the sweeps) | ms, per-GPU efficiency serial | hidden by the overlap, ms | ms, per-GPU efficiency overlapped |")
        print("|---|---|---|---|---|---|---|")
        for name, gen in gens:
            if not single.get(name):
                continue
            t1, per_sweep = single[name]
            t1 *= 2.0 ** (nl - 30)
            per_sweep *= 2.0 ** (nl - 30)
            for world in (2, 4, 8):
                g = world.bit_length() - 1
                n = nl + g
                m = plans[(name, world)]
                ov = sharded.debug_overlap(n, 0, world, gen(n), bits, P)["remaps"]
                comm = m["exchanges"] * m["exchange_ms"] + m["packs_from_row_positions"] * m["pack_ms"]
                hid = sum(hidden(m["exchange_ms"], per_sweep, r["before"], r["after"]) for r in ov)
                nb, na = sum(r["before"] for r in ov), sum(r["after"] for r in ov)
                print(f"| {name} | {world} | {nb} / {na} of {len(ov)} | {comm:.0f} | {t1 + comm:.0f} ms, {100 * t1 / (t1 + comm):.0f} % | {hid:.0f} | "
                      f"{t1 + comm - hid:.0f} ms, {100 * t1 / (t1 + comm - hid):.0f} % |")


if __name__ == "__main__":
    main()
