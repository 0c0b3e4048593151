"""Register use of a circuit's first run-time-compiled tile segment, without a GPU: the generated source
(qip_hip_debug_tile_jit) compiled offline with the run-time flags and -Rpass-analysis=kernel-resource-usage.
    python tools/jit_segment_resources.py [n = 30] [c2,c4,grover,qft] [modes: 1|4|64, 2|4|64|128 ...]
mode bits: 0-1 tile, 4 relabel, 16 wide tiles, 64 numbers as kernel data, 128 merged diagonal runs, 256 register pins, 512 dense-3 gates written out."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustqip_amd import circuits  # noqa: E402
from rustqip_amd.ops import debug_tile_jit  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    names = (sys.argv[2] if len(sys.argv) > 2 else "c2,c4,grover,qft").split(",")
    modes = [int(m) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1 | 4 | 64, 2 | 4 | 64 | 128]
    cases = {"c2": circuits.c2_random_circuit(n, 256, seed=28), "c4": circuits.c4_clifford_t(n, 256, seed=32),
             "grover": circuits.c5_grover_iteration(n), "groverk3": circuits.c5_grover_iteration(n, dense_k3=True), "qft": circuits.c3_qft(n)}
    for name in names:
        for mode in modes:
            r = debug_tile_jit(n, cases[name], mode)
            src = r["first_source"]
            src = src if isinstance(src, str) else src.decode()
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "seg.hip")
                open(path, "w").write(src)
                cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                       "-I" + os.path.join(ROOT, "rustqip_amd", "csrc"), "-c", path, "-o", os.path.join(d, "seg.o"),
                       "-Rpass-analysis=kernel-resource-usage"]
                out = subprocess.run(cmd, capture_output=True, text=True).stderr
            get = lambda key: (re.search(key + r":\s*(\d+)", out) or [None, "?"])[1]  # noqa: E731
            scratch, occ = get(r"ScratchSize \[bytes/lane\]"), get(r"Occupancy \[waves/SIMD\]")
            print("%-8s mode %3d: %d segments, first: VGPRs %s, scratch %s B/lane, VGPR spills %s, SGPR spills %s, occupancy %s"
                  % (name, mode, r["segments"], get("VGPRs"), scratch, get("VGPRs Spill"), get("SGPRs Spill"), occ))

if __name__ == "__main__":
    main()
