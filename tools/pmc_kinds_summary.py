"""Per-gate-kind instruction counts from tools/pmc_kinds.sh (gpurun_out/pmc_kinds/k_counter_collection.csv)."""
import collections
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_kinds/k_counter_collection.csv"
rows = collections.defaultdict(dict)
meta = {}
for r in csv.DictReader(open(path)):
    k = int(r["Dispatch_Id"])
    rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[k] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = ["20H", "20X", "20Rz", "20T", "20Y", "20dense", "20CNOT", "20CP", "2H"]
ks = [k for k in sorted(rows) if "k_tile_passes" in meta[k][0]]
i = 0
for mode in (1, 2):
    for nm in names:
        k = ks[i + 1]  # second launch of each case (the timed one)
        i += 2
        c = rows[k]
        w = c["SQ_WAVES"]
        print(f"tile={mode} {nm:8s} us={meta[k][1] / 1e3:6.0f}  VALU/wave={c['SQ_INSTS_VALU'] / w:6.0f}  SALU/wave={c['SQ_INSTS_SALU'] / w:6.0f}"
              f"  LDS/wave={c['SQ_INSTS_LDS'] / w:4.0f}  wave_life_cycles={4 * c['SQ_WAVE_CYCLES'] / w:7.0f}"
              f"  valu_busy={100 * c['SQ_INSTS_VALU'] * 4 / 1024 / (c['GRBM_GUI_ACTIVE'] / 8):3.0f}%")
