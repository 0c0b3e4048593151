#!/usr/bin/env python
"""What a `calculate_state` caller pays for the gate loop, in a process of its own: HipBuilder's run loop (rustqip_amd/builder.py:
one apply_ops batch per run of gates, options tile = 1 + tile_relabel = 1, a fresh state handle) on the configs[1] circuit
(n H gates, then 256 gates 3/4 single-qubit + 1/4 CNOT, seed 28), timed from the first gate to the device's
completion — without the state allocation before and the download after, which the reference pays in host memory as well.

  python tools/builder_one_shot.py [n]  ->  one JSON line {"ms": ..., "jit": {...}}

bench.py runs this as the "second process" of its `builder` section: a warm disk cache makes the same call take the compiled
wide sweeps (option tile_auto), a cold one the interpreter."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import rustqip_amd as q  # noqa: E402
from rustqip_amd import _ffi, circuits  # noqa: E402
from rustqip_amd.builder import HipBuilder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
b = HipBuilder()
prep = circuits.h_layer(n)
ops = circuits.c2_random_circuit(n, 256, seed=28)
state = q.HipState(n, np.complex128)
state.set_option("tile", b.tile)
state.set_option("tile_relabel", b.tile_relabel)
state.init_basis(0)
state.set_option("tile_auto", 0)  # (no plan lookup, no helper processes for the untimed preparation: they would still be starting up during the timed call)
state.apply_ops(prep)  # (the H layer that makes the state dense: part of configs[1]'s preparation, not of its timed 256 gates)
state.set_option("tile_auto", 1)
state.sync()
c0 = _ffi.jit_counters()
t0 = time.perf_counter()
state.apply_ops(ops)   # the run loop's one batch (builder.rs:423-517: no measurement in this circuit)
state.sync()
ms = 1e3 * (time.perf_counter() - t0)
c1 = _ffi.jit_counters()
norm = state.norm_sqr()
state.close()
print(json.dumps({"n": n, "gates": len(ops), "ms": ms, "norm_sqr": norm, "compiled_sweeps": c1["kernels_resident_total"] > c0["kernels_resident_total"],
                  "jit": {k: c1[k] - c0[k] if isinstance(c1[k], (int, float)) and k not in ("procs", "disk_cache") else c1[k] for k in c1}}))
