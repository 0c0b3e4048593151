#!/usr/bin/env python
"""Selector bits inside a 128-B line: per-lane predicate over whole lines vs removed from the grid (partial lines).

  python tools/exp_linebits.py [n]
For line_bits = 3 (default: bits 0..2 stay in the grid), 2, 1, 0: time phase gates and CNOTs whose selector sits on
bit 0 / 1 / 2 / 3.  GB/s uses the algorithmic bytes (half of the vector for these ops)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    reps = 6
    mid = n // 2
    cases = []
    for b in (0, 1, 2, 3):
        qb = n - 1 - b
        cases.append((f"T on bit {b}", q.make_matrix_op([qb], circuits.T)))
        cases.append((f"CNOT c=bit {b} t=n/2", q.make_control_op([qb], q.make_matrix_op([mid], circuits.X))))
        cases.append((f"CNOT c=bit {b} t=bit 4 (xlane)", q.make_control_op([qb], q.make_matrix_op([n - 1 - 4], circuits.X))))
        cases.append((f"CH c=bit {b} t=n/2", q.make_control_op([qb], q.make_matrix_op([mid], circuits.H))))
    cases.append(("Toffoli c=bits 1,2 t=n/2", q.make_control_op([n - 2, n - 3], q.make_matrix_op([mid], circuits.X))))
    cases.append(("CCZ bits 0,2,n/2", q.make_control_op([n - 1, n - 3], q.make_matrix_op([mid], circuits.Z))))
    print(f"| op (n={n}, f64) | " + " | ".join(f"line_bits={lb}: ms (GB/s)" for lb in (3, 2, 1, 0)) + " |")
    print("|---|---|---|---|---|")
    with q.HipState(n) as st:
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
        for name, op in cases:
            row = []
            for lb in (3, 2, 1, 0):
                q.set_global_option("line_bits", lb)
                comp = st.compile_ops([op] * reps)
                st.apply_compiled(st.compile_ops([op]))
                st.sync()
                t0 = time.perf_counter()
                st.apply_compiled(comp)
                st.sync()
                dt = (time.perf_counter() - t0) / reps
                by = q.algorithmic_bytes(n, op, 0)
                row.append(f"{dt*1e3:.3f} ({by/dt/1e9:.0f})")
            q.set_global_option("line_bits", 3)
            print(f"| {name} | " + " | ".join(row) + " |")
        print(f"\nnorm after all ops: {st.norm_sqr():.15f}")


if __name__ == "__main__":
    main()
