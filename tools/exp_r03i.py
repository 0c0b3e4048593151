"""H / X sweep over every target qubit: dedicated kernels (single_via_tile = 2) vs one-op tile sweeps (3).  n = 30."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rustqip_amd as q
from rustqip_amd import circuits
n = 30
with q.HipState(n) as st:
    st.init_basis(0)
    st.apply_ops(circuits.h_layer(n))
    for mode in (2, 3):
        q.set_global_option("single_via_tile", mode)
        for name, m in (("H", circuits.H), ("X", circuits.X)):
            row = []
            for tq in range(n):
                op = st.compile_ops([q.make_matrix_op([tq], m)] * 3)
                st.apply_compiled(op); st.sync()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); st.apply_compiled(op); st.sync(); ts.append((time.perf_counter() - t0) / 3)
                row.append(round(32.0 * 2**n / min(ts) / 1e9))
            print(mode, name, "min", min(row), "median", int(np.median(row)), row, flush=True)
    q.set_global_option("single_via_tile", 2)
