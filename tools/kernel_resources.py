#!/usr/bin/env python
"""Per-kernel register / scratch / LDS usage of the HIP extension, from hipcc's -Rpass-analysis=kernel-resource-usage.

  python tools/kernel_resources.py [substring ...]     (cross-compiles for gfx950; no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[: len(names)]
    except FileNotFoundError:
        return names


def main():
    res = "/tmp/qip_kernel_resources.txt"
    if not (len(sys.argv) > 1 and sys.argv[1] == "--cached" and os.path.exists(res)):
        with open(res, "w") as f:
            for unit in ("qip_launch", "qip_circuit", "qip_measure", "qip_dist"):  # the translation units that launch kernels
                cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                       "-c", "-o", "/tmp/qip_res.o", os.path.join(ROOT, "rustqip_amd", "csrc", unit + ".hip"),
                       "-Rpass-analysis=kernel-resource-usage"]
                subprocess.run(cmd, stderr=f, check=True)
    want = [a for a in sys.argv[1:] if not a.startswith("--")]
    txt = open(res).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((b.split("\n")[0].strip(), g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    names = demangle([r[0] for r in rows])
    print("vgpr agpr sgpr scratch occ lds  kernel")
    for r, nm in zip(rows, names):
        if not want or any(w in nm for w in want):
            print("%4d %4d %4d %6d %3d %5d  %s" % (r[1], r[2], r[3], r[4], r[5], r[6], nm[:160]))


if __name__ == "__main__":
    main()
