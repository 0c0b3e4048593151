#!/usr/bin/env python
"""Time the one-sweep bit permutation (k_permute_bits) on a resident 2^n state: GB/s = 2 * bytes of the vector / time.

  python tools/bench_permute.py [n] [f32]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
    amp = 8.0 if f32 else 16.0
    rng = np.random.default_rng(1)
    ident = list(range(n))

    def transp(*pairs):
        pi = list(ident)
        for a, b in pairs:
            pi[a], pi[b] = pi[b], pi[a]
        return pi

    top3 = [n - 3, n - 2, n - 1]
    cases = [
        ("bit reversal (QFT's closing swaps)", ident[::-1]),
        ("one transposition, bits 29 <-> 12 (high, high)", transp((n - 1, 12))),
        ("one transposition, bits 29 <-> 0", transp((n - 1, 0))),
        ("two transpositions (29 0)(15 1) [Swap(2) mixed]", transp((n - 1, 0), (15, 1))),
        ("three transpositions (29 0)(15 1)(22 2) [Swap(3)]", transp((n - 1, 0), (15, 1), (22, 2))),
        ("rotation by one (every bit moves)", list(range(1, n)) + [0]),
        ("random permutation of all bits", [int(v) for v in rng.permutation(n)]),
        ("random permutation of bits >= 6 (rows stay rows)", list(range(6)) + [6 + int(v) for v in rng.permutation(n - 6)]),
        ("multi-GPU pack: bits 10, 17, 23 to the top", [b for b in range(n) if b not in (10, 17, 23)] + [10, 17, 23]),
        ("multi-GPU pack: bits 1, 4, 20 to the top", [b for b in range(n) if b not in (1, 4, 20)] + [1, 4, 20]),
    ]
    print(f"| bit permutation (n={n}, Complex<{'f32' if f32 else 'f64'}>, out of place) | ms | GB/s | % of 8 TB/s |\n|---|---|---|---|")
    with q.HipState(n, np.complex64 if f32 else np.complex128) as st:
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n) + [q.make_matrix_op([t], circuits.rz(0.1 + 0.37 * t)) for t in range(n)])
        for name, pi in cases:
            st.permute_bits(pi)
            st.sync()
            reps = 6
            t0 = time.perf_counter()
            for _ in range(reps):
                st.permute_bits(pi)
            st.sync()
            dt = (time.perf_counter() - t0) / reps
            by = 2 * amp * 2**n
            print(f"| {name} | {dt*1e3:.3f} | {by/dt/1e9:.0f} | {100*by/dt/1e9/8000:.1f} |", flush=True)
        print(f"\nnorm: {st.norm_sqr():.15f}")


if __name__ == "__main__":
    main()
