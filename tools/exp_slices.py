"""What cutting a tile sweep into parts costs by itself (global option "debug_slice_sweeps"): the sliced launches of the overlapped
exchange, on one GPU, without any exchange.  Equality of the results at n = 24, then times at n = 30 (median of 5).
    python tools/exp_slices.py > gpurun_out/.../slices.jsonl"""
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402

MODES = {"interpreter": {"tile": 1}, "compiled 11-bit": {"tile": 1, "tile_jit": 1}, "compiled wide": {"tile": 1, "tile_jit": 1, "tile_wide": 1},
         "compiled wide, 1e-12 mode": {"tile": 2, "tile_jit": 1, "tile_wide": 1, "tile_fma": 1, "tile_merge": 1}}


def run(n, ops, opts, slices, reps, x=None):
    q.set_global_option("debug_slice_sweeps", slices)
    with q.HipState(n) as st:
        for k, v in opts.items():
            st.set_option(k, v)
        if x is not None:
            st.upload(x)
        else:
            st.init_basis(0)
            st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
        cc = st.compile_ops(ops)
        st.set_option("profile", 1)
        st.apply_compiled(cc)
        st.sync()
        prof = st.profile()
        launches = prof.get("tile_sweep_parts", {}).get("launches", 0)  # parts of sweeps that ran in slices
        st.set_option("profile", 0)
        ts = []
        for _ in range(reps):
            st.sync()
            t = time.perf_counter()
            st.apply_compiled(cc)
            st.sync()
            ts.append(time.perf_counter() - t)
        out = st.download() if x is not None else None
    q.set_global_option("debug_slice_sweeps", 0)
    return (statistics.median(ts) if ts else None), launches, out


def main():
    n = 24
    x = circuits.random_state(n, seed=3)
    for cname, ops in (("c2", circuits.c2_random_circuit(n, 128, seed=28)), ("qft", circuits.c3_qft(n)[:200])):
        for mname, opts in MODES.items():
            base = run(n, ops, opts, 0, 0, x)
            for P in (2, 4, 8):
                got = run(n, ops, opts, P, 0, x)
                assert np.array_equal(base[2], got[2]), (cname, mname, P)
                assert base[1] == 0 and got[1] >= 2 * P, (cname, mname, P, base[1], got[1])  # sweeps really ran in parts
    print(json.dumps({"equal": "sliced sweeps (2 / 4 / 8 parts) give the same amplitudes bit for bit at n = 24, every mode"}), flush=True)
    n = 30
    for cname, ops in (("c2", circuits.c2_random_circuit(n, 256, seed=28)), ("c4", circuits.c4_clifford_t(n, 256, seed=32))):
        for mname, opts in MODES.items():
            for P in (0, 4):
                t, launches, _ = run(n, ops, opts, P, 5)
                print(json.dumps({"circuit": cname, "n": n, "mode": mname, "parts": P or 1, "sliced_parts_launched": launches, "ms": round(1e3 * t, 2)}), flush=True)


if __name__ == "__main__":
    main()
