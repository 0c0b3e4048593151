#!/usr/bin/env python
"""Launch-bound regime: the reference's own bench shapes (qip/benches/state_bench.rs) are small states.
Prints gates/s for one op repeated 2000 times at n = 8..24 (one FFI crossing, async launches)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def main():
    reps = 2000
    print("| n | op | eager us/gate | eager gates/s | hipGraph us/gate | hipGraph gates/s | algorithmic GB/s (graph) |\n|---|---|---|---|---|---|---|")
    for n in (8, 12, 16, 20, 22, 24):
        for name, op in (("H on qubit 0 (state_bench.rs:141-155 shape)", q.make_matrix_op([0], circuits.H)),
                         ("CNOT", q.make_control_op([0], q.make_matrix_op([n - 1], circuits.X))),
                         ("random C2 circuit", None)):
            with q.HipState(n) as st:
                st.init_basis(0)
                st.apply_ops(circuits.h_layer(n))
                ops = circuits.c2_random_circuit(n, reps, seed=1) if op is None else [op] * reps
                comp = st.compile_ops(ops)
                st.apply_compiled(comp)
                st.sync()
                t = time.perf_counter()
                st.apply_compiled(comp)
                st.sync()
                dt = time.perf_counter() - t
                by = sum(q.algorithmic_bytes(n, o) for o in ops[:64]) / 64 * reps
                prog = st.compile_program(ops)
                prog.run()
                st.sync()
                t = time.perf_counter()
                prog.run()
                st.sync()
                dg = time.perf_counter() - t
                tag = "" if prog.is_graph else " (eager fallback)"
                prog.close()
                extra = ""
                if n >= 11 and op is None:
                    st.set_option("tile", 1)  # bit-identical multi-gate sweeps, then the same inside a graph
                    ct = st.compile_ops(ops)
                    st.apply_compiled(ct)
                    st.sync()
                    t = time.perf_counter()
                    st.apply_compiled(ct)
                    st.sync()
                    dtile = time.perf_counter() - t
                    prog = st.compile_program(ops)
                    prog.run()
                    st.sync()
                    t = time.perf_counter()
                    prog.run()
                    st.sync()
                    dtg = time.perf_counter() - t
                    extra = f" tile=1: {reps/dtile:.0f} gates/s eager, {reps/dtg:.0f} gates/s as a hipGraph"
                    prog.close()
                    st.set_option("tile", 0)
                print(f"| {n} | {name} | {1e6*dt/reps:.2f} | {reps/dt:.0f} | {1e6*dg/reps:.2f}{tag} | {reps/dg:.0f} | {by/dg/1e9:.1f} |{extra}")


if __name__ == "__main__":
    main()
