"""Probe what bounds one LDS-resident tile sweep: time segments that differ only in which index bits the tile
spans (contiguous vs scattered rows) and in how many gates ride along.   python tools/probe_tile.py [n]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    H = circuits.H
    bit = lambda b: n - 1 - b  # qubit index of index bit b
    cases = {
        "2H bits 6,7 (contiguous 32 KiB tile)": [6, 7],
        "2H bits 0,1 (pads 6..10)": [0, 1],
        "2H bits %d,%d (rows 8 GiB apart)" % (n - 2, n - 1): [n - 2, n - 1],
        "5H bits 6..10": [6, 7, 8, 9, 10],
        "5H bits top5": list(range(n - 5, n)),
        "5H bits 12,15,18,21,24": [12, 15, 18, 21, 24],
        "20H bits 6..10 x4": [6, 7, 8, 9, 10] * 4,
        "20H bits top5 x4": list(range(n - 5, n)) * 4,
        "20H bits 0..4 x4": [0, 1, 2, 3, 4] * 4,
    }
    if len(sys.argv) > 2 and sys.argv[2] == "kinds":
        # one gate kind per sweep, targets spread over lane / wave / register bits (PMC: instructions per gate)
        import cmath
        bits20 = [0, 3, 6, 7, 8, 9, 10, 2, 5, 8] * 2
        X = [0, 1, 1, 0]
        rz = [cmath.rect(1, -0.3), 0, 0, cmath.rect(1, 0.3)]
        tg = [1, 0, 0, cmath.rect(1, 0.785398)]
        yy = [0, -1j, 1j, 0]
        dense = [0.6, 0.8j, 0.8j, 0.6]
        kinds = {
            "20H": [q.make_matrix_op([bit(b)], H) for b in bits20],
            "20X": [q.make_matrix_op([bit(b)], X) for b in bits20],
            "20Rz": [q.make_matrix_op([bit(b)], rz) for b in bits20],
            "20T": [q.make_matrix_op([bit(b)], tg) for b in bits20],
            "20Y": [q.make_matrix_op([bit(b)], yy) for b in bits20],
            "20dense": [q.make_matrix_op([bit(b)], dense) for b in bits20],
            "20CNOT": [q.make_control_op([bit(bits20[(i + 3) % 20])], q.make_matrix_op([bit(b)], X))
                       for i, b in enumerate(bits20) if bits20[(i + 3) % 20] != b],
            "20CP": [q.make_control_op([bit(bits20[(i + 3) % 20])], q.make_matrix_op([bit(b)], tg))
                     for i, b in enumerate(bits20) if bits20[(i + 3) % 20] != b],
            "2H": [q.make_matrix_op([bit(b)], H) for b in (6, 7)],
        }
        with q.HipState(n) as st:
            st.init_basis(0)
            st.apply_ops(circuits.h_layer(n))
            for mode in (1, 2):
                st.set_option("tile", mode)
                for name, ops in kinds.items():
                    cops = st.compile_ops(ops)
                    st.apply_compiled(cops)
                    st.sync()
                    t0 = time.perf_counter()
                    st.apply_compiled(cops)
                    st.sync()
                    print(json.dumps({"tile": mode, "case": name, "gates": len(ops), "ms": round(1e3 * (time.perf_counter() - t0), 3)}), flush=True)
        return
    with q.HipState(n) as st:
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))
        for mode in (1, 2):
            st.set_option("tile", mode)
            for name, bits in cases.items():
                ops = [q.make_matrix_op([bit(b)], H) for b in bits]
                cops = st.compile_ops(ops)
                st.set_option("profile", 1)
                st.profile_reset()
                st.apply_compiled(cops)
                st.sync()
                sweeps = sum(v["launches"] for v in st.profile().values())
                st.set_option("profile", 0)
                best = 1e9
                for _ in range(3):
                    st.sync()
                    t0 = time.perf_counter()
                    st.apply_compiled(cops)
                    st.sync()
                    best = min(best, time.perf_counter() - t0)
                print(json.dumps({"tile": mode, "case": name, "sweeps": sweeps, "ms": round(1e3 * best, 3),
                                  "GBps_per_sweep": round(32 * 2.0**n * sweeps / best / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
