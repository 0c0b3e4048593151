"""Register use of EVERY run-time-compiled tile segment of a circuit's plan, without a GPU (r5; r4 looked at the first segment only):
the generated sources (qip_hip_debug_tile_jit with QIP_HIP_JIT_DUMP_DIR) compiled offline with the run-time flags and
-Rpass-analysis=kernel-resource-usage.

    python tools/jit_segment_resources.py [n = 30] [c2,c4,grover,groverk3,qft] [modes: 1|4|64, ...] [--md]

mode bits: 0-1 tile, 4 relabel, 16 wide tiles, 32 fused multiply-adds, 64 numbers as kernel data, 128 merged diagonal runs,
256 register pins, 512 dense-3 gates written out."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEYS = (("VGPRs", r"VGPRs"), ("AGPRs", r"AGPRs"), ("SGPRs", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("vspill", r"VGPRs Spill"), ("sspill", r"SGPRs Spill"), ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]"))


def resources(path, fma):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=" + ("fast" if fma else "off"),
               "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "rustqip_amd", "csrc"), "-c", path, "-o", os.path.join(d, "seg.o"),
               "-Rpass-analysis=kernel-resource-usage"]
        out = subprocess.run(cmd, capture_output=True, text=True).stderr
    # the remark block of the segment's own kernel (the header's templates are not instantiated)
    blk = out[out.find("Function Name: qip_segment"):] if "Function Name: qip_segment" in out else out
    res = {}
    for key, pat in KEYS:
        m = re.search(pat + r":\s*(\d+)", blk)
        res[key] = int(m.group(1)) if m else -1
    src = open(path).read()
    res["gates"] = src.count("// gate ") if "// gate " in src else -1
    res["src_kb"] = len(src) // 1024
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = "--md" in sys.argv
    n = int(args[0]) if len(args) > 0 else 30
    names = (args[1] if len(args) > 1 else "c2,c4,grover,groverk3,qft").split(",")
    modes = [int(m) for m in args[2].split(",")] if len(args) > 2 else [1 | 64, 1 | 16 | 64 | 256, 1 | 4 | 16 | 64 | 256, 2 | 16 | 32 | 64 | 128 | 256]
    if md:
        print("| circuit | mode | segment | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD | LDS B |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name in names:
        for mode in modes:
            with tempfile.TemporaryDirectory() as dump:
                code = ("import sys; sys.path.insert(0, %r)\nfrom rustqip_amd import circuits\nfrom rustqip_amd.ops import debug_tile_jit\n"
                        "n = %d\ncases = {'c2': lambda: circuits.c2_random_circuit(n, 256, seed=28), 'c4': lambda: circuits.c4_clifford_t(n, 256, seed=32),"
                        "'grover': lambda: circuits.c5_grover_iteration(n), 'groverk3': lambda: circuits.c5_grover_iteration(n, dense_k3=True),"
                        "'qft': lambda: circuits.c3_qft(n)}\nr = debug_tile_jit(n, cases[%r](), %d)\nprint(r['segments'])\n") % (ROOT, n, name, mode)
                env = dict(os.environ, QIP_HIP_JIT_DUMP_DIR=dump)
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
                if p.returncode != 0:
                    print(name, mode, "FAILED", p.stderr[-500:])
                    continue
                files = sorted(os.path.join(dump, f) for f in os.listdir(dump))
                with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
                    rows = list(ex.map(lambda f: resources(f, f.endswith("_fma.hip")), files))
            worst = {k: max(r[k] for r in rows) for k, _ in KEYS}
            if md:
                for i, r in enumerate(rows):
                    print(f"| {name} | {mode} | {i + 1}/{len(rows)} | {r['VGPRs']} | {r['AGPRs']} | {r['SGPRs']} | {r['scratch']} | {r['vspill']} | {r['sspill']} | {r['occ']} | {r['lds']} |")
            else:
                print("%-8s mode %4d: %2d segments; max VGPRs %d AGPRs %d, scratch <= %d B/lane, VGPR spills <= %d, SGPR spills <= %d, occupancy >= %d; segments with scratch: %d"
                      % (name, mode, len(rows), worst["VGPRs"], worst["AGPRs"], worst["scratch"], worst["vspill"], worst["sspill"],
                         min(r["occ"] for r in rows), sum(1 for r in rows if r["scratch"] > 0)), flush=True)


if __name__ == "__main__":
    main()
