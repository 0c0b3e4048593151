#!/usr/bin/env python
"""A/B of the global option tile_wide_pin on one GPU (seconds): bit identity at n = 16, sweep time of a Clifford+T prefix at n = 30."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def run(n, ops, pin, reps=0):
    q.set_global_option("tile_wide_pin", pin)
    with q.HipState(n) as st:
        for k, v in (("tile", 1), ("tile_jit", 1), ("tile_wide", 1)):
            st.set_option(k, v)
        if reps == 0:
            st.upload(circuits.random_state(n, seed=3))
            st.apply_ops(ops)
            return st.download()
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))
        c = st.compile_ops(ops)
        st.apply_compiled(c)
        st.sync()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            st.apply_compiled(c)
            st.sync()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2] * 1e3


name = sys.argv[2] if len(sys.argv) > 2 else "c4"
if name == "c4":
    a = run(16, circuits.c4_clifford_t(16, 48, seed=32), 0)
    b = run(16, circuits.c4_clifford_t(16, 48, seed=32), 1)
    print("n=16 bit-identical:", bool(np.array_equal(a, b)))
full = circuits.c4_clifford_t(30, 256, seed=32) if name == "c4" else circuits.c2_random_circuit(30, 256, seed=28)
ops = full[:int(sys.argv[1]) if len(sys.argv) > 1 else 72]
for pin in (0, 1):
    print("n=30 %s prefix of %d gates, tile = 1 wide, pin = %d: %.2f ms (median of 5)" % (name, len(ops), pin, run(30, ops, pin, 5)))
q.set_global_option("tile_wide_pin", 1)
