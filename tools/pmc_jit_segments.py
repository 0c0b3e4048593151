#!/usr/bin/env python
"""HBM traffic of every run-time-compiled tile segment of a plan, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs)
over `tools/bench_tile.py <n> 1 <circuits> <mode>` (r5; VERDICT r4: scratch traffic of the wide legs was unmeasured).

  python tools/pmc_jit_segments.py <fetch_dir> <write_dir> <label>      -> markdown on stdout

Every launch of `qip_segment` is one row of rocprofv3's counter CSV, in dispatch order; bench_tile applies the circuit (profile pass +
warm-up + reps) several times, so the k-th segment of the plan is dispatch k modulo the plan length (given by the sweeps count bench_tile
prints; here: detected as the period of the FETCH sequence).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the guide's gfx950
correction, calibrated in profiles/r0*_pmc_traffic.md); the algorithmic bytes of a sweep are 32 * 2^n."""
import collections
import csv
import glob
import os
import sys


def load(d, counter):
    rows = []
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
        if r["Counter_Name"] == counter and "qip_segment" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    return rows


def period(seq):
    for p in range(1, len(seq) // 2 + 1):
        if len(seq) % p == 0 and all(abs(seq[i] - seq[i % p]) <= 0.02 * max(seq[i % p], 1) for i in range(len(seq))):
            return p
    return len(seq)


def main():
    fetch_dir, write_dir, label = sys.argv[1:4]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    alg = 32.0 * 2 ** n
    f, w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    # bench_tile.py <n> 1 ... applies the circuit exactly twice (the profiled pass, one timed repetition): the plan is half the launches
    # (an explicit sixth argument overrides; the automatic period search is only the fallback for other drivers)
    apps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    p = len(f) // apps if apps and len(f) % apps == 0 else period([x[1] for x in f])
    print(f"## {label}: {len(f)} launches of run-time-compiled segments, plan length {p}\n")
    print("| segment | launches | avg ms (under the profiler) | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes (2 F + W) x 1024 | x algorithmic (32 * 2^n) |")
    print("|---|---|---|---|---|---|---|")
    worst = 0.0
    for k in range(p):
        fk = [x for i, x in enumerate(f) if i % p == k]
        wk = [x for i, x in enumerate(w) if i % p == k]
        fm = sum(x[1] for x in fk) / len(fk)
        wm = sum(x[1] for x in wk) / max(len(wk), 1)
        ms = sum(x[2] for x in fk) / len(fk) / 1e6
        hbm = (2 * fm + wm) * 1024
        worst = max(worst, hbm / alg)
        print(f"| {k + 1}/{p} | {len(fk)} | {ms:.2f} | {fm:.0f} | {wm:.0f} | {hbm:.4e} | {hbm / alg:.4f} |")
    print(f"\nlargest ratio: {worst:.4f}\n")


if __name__ == "__main__":
    main()
