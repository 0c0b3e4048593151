"""Summarise gpurun_out/pmc_tile{A,B}/*_counter_collection.csv per kernel (mean per launch)."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for f in sorted(glob.glob(root + "/pmc_tile*/*_counter_collection.csv")):
    rows = collections.defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[k] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.Counter()
    for k, c in rows.items():
        name = meta[k][0]
        if "tile" not in name and "qip_segment" not in name:  # (qip_segment: the run-time-compiled tile sweeps)
            continue
        key = name.split("(")[0][-40:]
        cnt[key] += 1
        dur[key] += meta[k][1]
        for cn, v in c.items():
            agg[key][cn] += v
    for key in agg:
        print(f"{key}  launches={cnt[key]}  avg_us={dur[key] / cnt[key] / 1e3:.0f}")
        for cn, v in sorted(agg[key].items()):
            print(f"    {cn:24s} {v / cnt[key]:.4g}")
