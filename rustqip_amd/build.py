"""Compile the HIP extension in-tree: rustqip_amd/csrc/*.hip -> rustqip_amd/lib/libqip_hip.so.

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the numerics
contract (no product may be fused into an add; see csrc/qip_kernels.h)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "qip_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "qip_kernels.h"), os.path.join(HERE, "csrc", "qip_dist.inc"),
        os.path.join(HERE, "..", "include", "qip_hip.h")]
OUT = os.path.join(HERE, "lib", "libqip_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass pairs f32 products into v_pk_mul_f32 / v_pk_add_f32, which need their operands in
# aligned register pairs: in the f32 tile-sweep kernel that cost 180 VGPRs (2 waves per SIMD, 8.7 ms per sweep) against 90
# without it; the kernels are bound by HBM or by instruction issue, never by f32 flops (profiles/r02_slp.md)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]
LIBS = ["-ldl"]  # librccl is dlopen-ed on first multi-GPU use (csrc/qip_dist.inc): no link-time dependency


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


EMBED = os.path.join(HERE, "csrc", "qip_kernels_embed.inc")


def write_embed() -> None:
    """qip_kernels.h as a C++ raw string literal: the library hands it to hiprtc when it compiles a tile segment at run
    time (option "tile_jit"), so the .so does not depend on the source tree."""
    with open(DEPS[1]) as f:
        text = f.read()
    assert ')QIPKSRC"' not in text
    body = 'R"QIPKSRC(' + text + ')QIPKSRC"\n'
    if not os.path.exists(EMBED) or open(EMBED).read() != body:
        with open(EMBED, "w") as f:
            f.write(body)


def build(force: bool = False) -> str:
    write_embed()
    if force or needs_build():
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = [HIPCC, *FLAGS, "-o", OUT, SRC, *LIBS]
        print("[rustqip_amd.build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
