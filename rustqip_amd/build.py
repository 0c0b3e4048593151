"""Compile the HIP extension in-tree: rustqip_amd/csrc/*.hip -> rustqip_amd/build/*.o -> rustqip_amd/lib/libqip_hip.so.

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the numerics
contract (no product may be fused into an add; see csrc/qip_kernels.h)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# one translation unit per concern (csrc/qip_internal.h lists them); compiled in parallel, linked into ONE library whose
# only exports are the C ABI (csrc/exports.map)
UNITS = ["qip_core", "qip_launch", "qip_tile_sched", "qip_circuit", "qip_host", "qip_measure", "qip_dist"]
HEADERS = [os.path.join(CSRC, h) for h in ("qip_kernels.h", "qip_internal.h", "qip_tile.h")] + [
    os.path.join(HERE, "..", "include", "qip_hip.h"), os.path.join(HERE, "..", "include", "qip_hip_debug.h")]
OBJDIR = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "lib", "libqip_hip.so")
JITC = os.path.join(HERE, "lib", "qip_jitc")
JITC_SRC = os.path.join(CSRC, "qip_jitc.c")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass pairs f32 products into v_pk_mul_f32 / v_pk_add_f32, which need their operands in
# aligned register pairs: in the f32 tile-sweep kernel that cost 180 VGPRs (2 waves per SIMD, 8.7 ms per sweep) against 90
# without it; the kernels are bound by HBM or by instruction issue, never by f32 flops (profiles/r02_slp.md)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]
# QIP_HIP_TUNING=1 at build time: the measured alternatives that the product build fixes at their defaults become options again
# (csrc/qip_core.hip, "tuning options"): what tools/bench_ops.py, bench_tile.py, bench_permute.py switch for A/B runs
if os.environ.get("QIP_HIP_TUNING"):
    FLAGS.append("-DQIP_HIP_TUNING")
LINK = ["--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
LIBS = ["-ldl"]  # librccl and libhiprtc are dlopen-ed on first use (qip_dist.hip, qip_circuit.hip): no link-time dependency

EMBED = os.path.join(CSRC, "qip_kernels_embed.inc")


def write_embed() -> None:
    """qip_kernels.h as a C++ raw string literal: the library hands it to hiprtc when it compiles a tile segment at run
    time (option "tile_jit"), so the .so does not depend on the source tree."""
    with open(HEADERS[0]) as f:
        text = f.read()
    assert ')QIPKSRC"' not in text
    body = 'R"QIPKSRC(' + text + ')QIPKSRC"\n'
    if not os.path.exists(EMBED) or open(EMBED).read() != body:
        with open(EMBED, "w") as f:
            f.write(body)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(OUT, [os.path.join(CSRC, u + ".hip") for u in UNITS] + HEADERS + [os.path.join(CSRC, "exports.map")]) or _stale(JITC, [JITC_SRC, OUT])


def build(force: bool = False) -> str:
    write_embed()
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    jobs = []
    for u in UNITS:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(OBJDIR, u + ".o")
        deps = [src] + HEADERS + ([EMBED] if u == "qip_circuit" else [])
        if force or _stale(obj, deps):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            print("[rustqip_amd.build]", " ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJDIR, u + ".o") for u in UNITS]
    if force or jobs or _stale(OUT, objs + [os.path.join(CSRC, "exports.map")]):
        cmd = [HIPCC, *LINK, "-o", OUT, *objs, *LIBS]
        print("[rustqip_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    # the run-time compiler's helper process (plain C over the C ABI; found by the library next to itself)
    if force or _stale(JITC, [JITC_SRC, OUT]):
        cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-o", JITC, JITC_SRC, "-L" + os.path.dirname(OUT), "-lqip_hip",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]
        print("[rustqip_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
