#!/usr/bin/env python
"""profiles/<tag>_real_p_rocprof.md from the rocprofv3 passes of tools/profile_real_p.sh: per real-P kernel, the average launch
duration (--kernel-trace --stats) and the HBM bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes, corrected
as the MI355X guide prescribes and tools/summarize_rocprof.py does (counters in KiB; FETCH_SIZE reports half of a wide coalesced
stream on gfx950: hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024), beside the algorithmic bytes of the shape.

  python tools/summarize_real_p.py r06 gpurun_out/r06_realp"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_real_groups_low|k_real_groups|k_gather_real)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else None


def main():
    tag, d = sys.argv[1], sys.argv[2]
    stats = {}
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "prof_stats", "*kernel_stats.csv"))[0])):
        k = short(r["Name"])
        if k:
            stats[k] = (int(r["Calls"]), float(r["AverageNs"]))

    def load(sub, counter):
        out = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k and r["Counter_Name"] == counter:
                    out[k].append(float(r["Counter_Value"]))
        return out

    fetch, write = load("prof_fetch", "FETCH_SIZE"), load("prof_write", "WRITE_SIZE")
    n = 28
    lines = [f"# {tag} — the real-P kernels under rocprofv3 (`tools/profile_real_p.sh`: n = {n}, three calls per shape, f64 and f32)", "",
             "Average launch duration from `--kernel-trace --stats`; HBM bytes per launch from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes",
             "(KiB counters; `hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024`, the gfx950 correction of `tools/summarize_rocprof.py`). Algorithmic bytes =",
             "`sizeof(P)` x 2^n x (2 overwrite, 3 accumulate). Template arguments: `k_real_groups<P, V, K, controls, swap, non-temporal>`,",
             "`k_real_groups_low<P, K, controls, swap, low bit, mode>`.", "",
             "| kernel | calls | avg launch us | algorithmic GB/s | % of 8 TB/s | HBM bytes / launch (PMC) | x algorithmic |", "|---|---|---|---|---|---|---|"]
    for k, (calls, avg) in sorted(stats.items(), key=lambda kv: kv[0]):
        eb = 8 if ("double" in k or "unsigned long" in k) else 4
        # the profile workload: the accumulating shape is the K = 1 op on qubit 0 (V > 1, non-temporal); everything else overwrites
        acc = bool(re.match(r"k_real_groups<(double, 2|float, 4), 1, 0, false, true>", k))
        alg = eb * (1 << n) * (3 if acc else 2)
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        hbm = (2 * f + w) * 1024
        lines.append(f"| `{k}` | {calls} | {avg / 1e3:.1f} | {alg / avg:.0f} | {alg / avg / 80:.1f} | {hbm:.4g} | {hbm / alg:.3f} |")
    out = os.path.join(ROOT, "profiles", f"{tag}_real_p_rocprof.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
