"""Longer seeded stress of the tile sweeps than the test suite runs: random circuits over every tileable gate
shape (and some that are not), tile = 1 must equal the gate-by-gate result under IEEE ==, tile = 2 to 1e-12.

    python tools/stress_tile.py [seeds] [gates]"""
import cmath
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402

S2 = 0.5 ** 0.5
G1 = {
    "X": [0, 1, 1, 0], "Y": [0, -1j, 1j, 0], "Z": [1, 0, 0, -1], "H": [S2, S2, S2, -S2], "S": [1, 0, 0, 1j],
    "T": [1, 0, 0, cmath.rect(1, 0.785398)], "Rz": [cmath.rect(1, -0.35), 0, 0, cmath.rect(1, 0.35)],
    "upper": [1, 1, 0, 1], "dense": [0.3 + 0.1j, -0.7j, 0.2, 0.9 - 0.4j], "ident": [1, 0, 0, 1],
}


def rand_unitary(k, rng):
    a = rng.standard_normal((1 << k, 1 << k)) + 1j * rng.standard_normal((1 << k, 1 << k))
    u, _ = np.linalg.qr(a)
    return u


def circuit(n, rng, gates):
    names = list(G1)
    ops = []
    for _ in range(gates):
        perm = [int(v) for v in rng.permutation(n)]
        shape = int(rng.integers(0, 10))
        nc = int(rng.integers(0, min(5, n - 3)))
        if shape <= 4:
            g = q.make_matrix_op([perm[0]], G1[names[int(rng.integers(0, len(names)))]])
            ops.append(q.make_control_op(perm[1:1 + nc], g) if nc and rng.integers(0, 2) else g)
        elif shape == 5:
            g = q.make_matrix_op([perm[0]], [1, 0, 0, cmath.rect(1, float(rng.uniform(0, 6.28)))])
            ops.append(q.make_control_op(perm[1:2 + nc], g))
        elif shape == 6:
            g = q.make_swap_op([perm[0]], [perm[1]])
            ops.append(q.make_control_op(perm[2:2 + nc], g) if nc else g)
        elif shape == 7:
            g = q.make_matrix_op(perm[:2], rand_unitary(2, rng).ravel())
            ops.append(q.make_control_op(perm[2:2 + min(nc, 3)], g) if nc else g)
        elif shape == 8:
            ops.append(q.make_matrix_op(perm[:3], rand_unitary(3, rng).ravel()))
        else:
            ops.append(q.make_swap_op(perm[:2], perm[2:4]))
    return ops


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    gates = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for seed in range(seeds):
        n = 11 + seed % 7
        rng = np.random.default_rng(1000 + seed)
        ops = circuit(n, rng, gates)
        x = circuits.random_state(n, seed=seed)
        res = {}
        for mode in (0, 1, 2):
            with q.HipState(n) as st:
                st.set_option("mfma", 0)
                st.set_option("tile", mode)
                st.upload(x)
                st.apply_ops(ops)
                res[mode] = st.download()
        ok1 = np.array_equal(res[1], res[0])
        err2 = float(np.max(np.abs(res[2] - res[0])))
        if not ok1 or err2 > 1e-12:
            bad += 1
            print(f"FAIL seed={seed} n={n}: tile1 equal={ok1} tile2 err={err2:.2e}", flush=True)
    print(f"stress: {seeds} circuits x {gates} gates, n = 11..17: {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
