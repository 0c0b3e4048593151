#!/usr/bin/env python
"""Turn rocprofv3 CSV output (gpurun_out/) into the committed summaries under profiles/.

  python tools/summarize_rocprof.py <round-tag> <stats_dir> <fetch_dir> <write_dir> [bench_json] [stats_extras_dir]

* <stats_dir>: `rocprofv3 --kernel-trace --stats` of `bench.py`      -> profiles/<tag>_kernel_stats.csv/.md
* <fetch_dir>/<write_dir>: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
                                                                     -> profiles/<tag>_pmc_traffic.md, profiles/pmc_traffic.json
HBM bytes are computed as the MI355X guide prescribes: counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of a wide coalesced stream (calibrated here on k_chunk_norms, a read-only sweep of
a known 16 GiB, and on the 16-GiB memset for WRITE_SIZE), so hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"void qipk::(k_[a-z0-9_]+)", name)
    k = m.group(1) if m else name.split("(")[0]
    return k  # (the library's profiling classes carry the kernels' own names: the tile sweeps are k_tile_passes)


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    bench_json = sys.argv[5] if len(sys.argv) > 5 else None
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats_csv = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0]
    shutil.copy(stats_csv, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats_csv)))
    by_cls = collections.OrderedDict()
    for r in rows:
        c = by_cls.setdefault(short(r["Name"]), {"calls": 0, "ns": 0})
        c["calls"] += int(r["Calls"])
        c["ns"] += int(r["TotalDurationNs"])
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats of `python bench.py --headline-only --steps 3 --warmup 1`",
             "(MI355X, n=30, Complex<f64>; the process also runs the 30 gates that prepare the resident product state and one k_chunk_norms)", "",
             "Per template instantiation (rocprofv3's own table):", "",
             "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| `{r['Name'].split('(')[0].replace('void ', '')}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | "
                     f"{int(r['TotalDurationNs'])/1e6:.1f} | {r['Percentage']} |")
    lines += ["", "Per kernel class (what bench.py's `kernels` / `roofline` objects aggregate):", "",
              "| class | calls | avg ms |", "|---|---|---|"]
    for k, v in by_cls.items():
        lines.append(f"| `{k}` | {v['calls']} | {v['ns']/v['calls']/1e6:.3f} |")
    if bench_json and os.path.exists(bench_json):
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        lines += ["", "bench.py line of the same (profiled) run — HIP-event averages to compare with the table above:", "",
                  "```json", json.dumps({k: b[k] for k in ("value", "ms_per_step", "roofline", "kernels") if k in b}, indent=1), "```"]
    open(os.path.join(prof, f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

    if len(sys.argv) > 6:  # the same command with every extras leg: per-instantiation table only
        ecsv = glob.glob(os.path.join(sys.argv[6], "*kernel_stats.csv"))[0]
        erows = list(csv.DictReader(open(ecsv)))
        el = [f"# {tag}: rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity` (extras included)",
              "(tile sweeps: `k_tile_passes` = the interpreter, `qip_segment` = run-time-compiled segments; fusion, QFT, Clifford+T, Grover legs)", "",
              "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
        for r in erows:
            el.append(f"| `{r['Name'].split('(')[0].replace('void ', '')}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | "
                      f"{int(r['TotalDurationNs'])/1e6:.1f} | {r['Percentage']} |")
        open(os.path.join(prof, f"{tag}_kernel_stats_with_extras.md"), "w").write("\n".join(el) + "\n")

    def load(d, counter):
        out = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
            if r["Counter_Name"] == counter:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return out

    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    traffic = {}
    lines = [f"# {tag}: HBM traffic per launch from rocprofv3 PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE)", "",
             "Counters are KiB. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts half of a wide coalesced read;",
             "calibration inside this very run: `k_chunk_norms` reads a known 16 GiB and reports 8 GiB; the 16-GiB memset",
             "reports WRITE_SIZE = 16 GiB exactly. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.", "",
             "| kernel class | launches | mean FETCH_SIZE KiB | mean WRITE_SIZE KiB | HBM bytes / launch (corrected) |", "|---|---|---|---|---|"]
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        hbm = (2 * f + w) * 1024
        lines.append(f"| `{k}` | {len(fetch.get(k, []))} | {f:.1f} | {w:.1f} | {hbm:.4e} |")
        if k.startswith("k_"):
            traffic[k] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib": f, "write_size_kib": w, "launches": len(fetch.get(k, []))}
    open(os.path.join(prof, f"{tag}_pmc_traffic.md"), "w").write("\n".join(lines) + "\n")
    traffic["_source"] = f"profiles/{tag}_pmc_traffic.md"
    # which kernels these figures describe: bench.py flags `traffic` as stale when csrc/qip_kernels.h has changed since
    import hashlib

    sha_file = os.path.join(os.path.dirname(os.path.normpath(stats_dir)), "kernels_sha16.txt")  # written on the GPU box by profile_round.sh
    if os.path.exists(sha_file):
        traffic["_kernels_sha16"] = open(sha_file).read().strip()
    else:
        traffic["_kernels_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "rustqip_amd", "csrc", "qip_kernels.h"), "rb").read()).hexdigest()[:16]
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
