"""Time the LDS-resident multi-gate sweeps (option "tile") on the configured circuits.

    python tools/bench_tile.py [n] [reps] [circuits, e.g. c2,qft] [modes, e.g. 1,2] [f32]      (QIP_TILE_JIT=1: run-time-compiled segments; QIP_TILE_RELABEL=1: qubit relabelling)

Prints one JSON line per (circuit, mode): sweeps launched, ms per run of the circuit, gates/s, and the
per-sweep HBM rate (each sweep reads and writes the vector once: 32 * 2^n bytes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def brickwork(n, layers, seed=7):
    """layers of random dense 2-qubit unitaries on neighbouring qubits (even / odd bonds alternate)"""
    import numpy as np

    rng = np.random.default_rng(seed)
    ops = []
    for layer in range(layers):
        for a in range(layer % 2, n - 1, 2):
            m = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
            u, _ = np.linalg.qr(m)
            ops.append(q.make_matrix_op([a, a + 1], u.ravel()))
    return ops


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cases = {
        "c2": circuits.c2_random_circuit(n, 256, seed=28),
        "c2x4": circuits.c2_random_circuit(n, 1024, seed=28),
        "c4": circuits.c4_clifford_t(n, 256, seed=32),
        "qft": circuits.c3_qft(n),
        "grover": circuits.c5_grover_iteration(n),
        "groverk3": circuits.c5_grover_iteration(n, dense_k3=True),
        "brick2q": brickwork(n, 4),
    }
    if len(sys.argv) > 3:
        cases = {k: v for k, v in cases.items() if k in sys.argv[3].split(",")}
    modes = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2]
    import numpy as np

    f32 = len(sys.argv) > 5 and sys.argv[5] == "f32"
    for key in ("tile_pad_from", "tile_wave_rule", "tile_remap", "tile_sched", "tile_row_split", "tile_row_split_f32", "tile_diag_runs",
                "tile_wide_dense3_inline", "tile_wide_pin", "jit_procs", "debug_slice_sweeps"):  # tuning aids (global options)
        if os.environ.get("QIP_" + key.upper()):
            q.set_global_option(key, int(os.environ["QIP_" + key.upper()]))
    tune = {k: os.environ.get("QIP_" + k.upper(), "") for k in ("tile_pad_from", "tile_wave_rule", "tile_remap", "tile_sched", "tile_row_split", "tile_row_split_f32",
                                                                "tile_diag_runs", "tile_wide_dense3_inline", "debug_slice_sweeps") if os.environ.get("QIP_" + k.upper())}
    with q.HipState(n, np.complex64 if f32 else np.complex128) as st:
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))
        for name, ops in cases.items():
            for mode, passes in [(m, 1) for m in modes]:
                st.set_option("tile", mode)
                try:
                    st.set_option("tile_passes", passes)
                except q.CircuitError:  # (a tuning build's option; the product build keeps passes = 1)
                    if passes != 1:
                        continue
                st.set_option("tile_jit", int(os.environ.get("QIP_TILE_JIT", "0")) if mode else 0)
                st.set_option("tile_relabel", int(os.environ.get("QIP_TILE_RELABEL", "0")) if mode else 0)
                st.set_option("tile_fma", int(os.environ.get("QIP_TILE_FMA", "0")) if mode else 0)
                st.set_option("tile_merge", int(os.environ.get("QIP_TILE_MERGE", "0")) if mode else 0)
                st.set_option("tile_wide", int(os.environ.get("QIP_TILE_WIDE", "0")) if mode else 0)
                for kv in filter(None, os.environ.get("QIP_STATE_OPTS", "").split(",")):  # e.g. QIP_STATE_OPTS=pair_floor=0
                    st.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                cops = st.compile_ops(ops)
                st.set_option("profile", 1)
                st.profile_reset()
                st.apply_compiled(cops)
                st.sync()
                sweeps = sum(v["launches"] for v in st.profile().values())
                st.set_option("profile", 0)
                best = 1e9
                for _ in range(reps):
                    st.sync()
                    t0 = time.perf_counter()
                    st.apply_compiled(cops)
                    st.sync()
                    best = min(best, time.perf_counter() - t0)
                print(json.dumps({"circuit": name, "n": n, "tile": mode, "gates": len(ops), "sweeps": sweeps,
                                  "ms": round(1e3 * best, 2), "gates_per_s": round(len(ops) / best, 1),
                                  "ms_per_sweep": round(1e3 * best / sweeps, 3), "dtype": "f32" if f32 else "f64", "jit": os.environ.get("QIP_TILE_JIT", "0"), "relabel": os.environ.get("QIP_TILE_RELABEL", "0"), "fma": os.environ.get("QIP_TILE_FMA", "0"), "merge": os.environ.get("QIP_TILE_MERGE", "0"), "wide": os.environ.get("QIP_TILE_WIDE", "0"),
                                  "tune": dict(tune, state_opts=os.environ.get("QIP_STATE_OPTS", "")), "norm": st.norm_sqr()}), flush=True)
        st.set_option("tile", 0)
        st.set_option("tile_jit", 0)
        st.set_option("tile_relabel", 0)
        st.set_option("tile_fma", 0)
        st.set_option("tile_merge", 0)
        st.set_option("tile_wide", 0)


if __name__ == "__main__":
    main()
