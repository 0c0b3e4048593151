#!/usr/bin/env python
"""stdin for `tools/tune_tile 30 5 probe2`: the tile shapes k_permute_bits uses today (R = 5 / 6) and split-row variants (lane bit 5 = index
position 11 on both sides) for the permutations of tools/bench_permute.py.  A line = tag TB s[0..TB-1] d[0..TB-1]."""
import sys

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ident = list(range(n))


def transp(*pairs):
    pi = list(ident)
    for a, b in pairs:
        pi[a], pi[b] = pi[b], pi[a]
    return pi


rng = np.random.default_rng(1)
cases = [
    ("copy", ident),
    ("reversal", ident[::-1]),
    ("transp_29_12", transp((n - 1, 12))),
    ("transp_29_0", transp((n - 1, 0))),
    ("two_transp", transp((n - 1, 0), (15, 1))),
    ("three_transp", transp((n - 1, 0), (15, 1), (22, 2))),
    ("rotation", list(range(1, n)) + [0]),
    ("random", [int(v) for v in rng.permutation(n)]),
    ("random_ge6", list(range(6)) + [6 + int(v) for v in rng.permutation(n - 6)]),
    ("pack_10_17_23", [b for b in range(n) if b not in (10, 17, 23)] + [10, 17, 23]),
    ("pack_1_4_20", [b for b in range(n) if b not in (1, 4, 20)] + [1, 4, 20]),
]


def tile(pi, R, TB, force=()):
    """destination tile positions (k_permute_bits' rule) and the matching source positions, in thread-bit order"""
    T = set(range(R)) | {b for b in range(n) if pi[b] < R}
    for f in force:  # positions that must be lane / tile bits on both sides
        T |= {f} | {b for b in range(n) if pi[b] == f}
    b = 0
    while len(T) < TB:
        if b not in T:
            T.add(b)
        b += 1
    if len(T) > TB:
        return None
    S = {pi[b] for b in T}
    first = list(range(R)) + [f for f in force]
    d = [p for p in first if p in T] + sorted(p for p in T if p not in first)
    s = [p for p in first if p in S] + sorted(p for p in S if p not in first)
    if len(d) != TB or len(s) != TB:
        return None
    return s, d


for name, pi in cases:
    for tag, R, TB, force in (("cur5", 5, 10, ()), ("cur6", 6, 12, ()), ("split11_tb11", 5, 11, (11,)), ("split11_tb12", 5, 12, (11,)),
                              ("rows5_tb11", 5, 11, ()), ("rows5_tb12", 5, 12, ())):
        t = tile(pi, R, TB, force)
        if t is None:
            continue
        s, d = t
        print(f"{name}.{tag} {TB} " + " ".join(map(str, s)) + " " + " ".join(map(str, d)))
