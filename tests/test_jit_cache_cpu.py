"""The run-time compiler's disk cache and helper processes (r5), without a GPU: hiprtc cross-compiles for gfx950 on the host, so
everything up to "the code object is in memory" runs here — what needs a device is only hipModuleLoadData.
Every case runs in a process of its own (the cache directory, the helper path and the counters are process-global)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
from rustqip_amd import circuits, _ffi
from rustqip_amd.ops import debug_tile_jit
import rustqip_amd as q
n, mode, procs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if procs >= 0:
    q.set_global_option("jit_procs", procs)
ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 96, seed=7)
t = time.time()
r = debug_tile_jit(n, ops, mode)
out = dict(_ffi.jit_counters())
out.update(segments=r["segments"], code_bytes=r["code_bytes"], sec=time.time() - t, dir=_ffi.lib.qip_hip_jit_cache_dir().decode())
print(json.dumps(out))
"""


def run(n, mode, procs, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, str(n), str(mode), str(procs)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def files(d):
    return sorted(f for f in os.listdir(d) if f.endswith(".co"))


def test_second_process_loads_from_disk_and_helpers_make_the_same_code(tmp_path):
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    n, mode = 16, 1 | 64  # tile = 1, numbers as kernel data: the default compiled form
    # helper processes (automatic count) fill directory a
    r1 = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": a})
    assert r1["dir"] == a and r1["disk_cache"] == 1
    assert r1["segments"] >= 3 and r1["compiled"] == r1["segments"] and r1["disk_hits"] == 0
    if r1["procs"] > 1:
        assert r1["compiled_by_helpers"] == r1["segments"] and 1 <= r1["helper_processes"] <= r1["procs"]
    assert len(files(a)) == r1["segments"] and not [f for f in os.listdir(a) if not f.endswith(".co")]  # no sources / temporaries left
    # a second process compiles nothing
    r2 = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": a})
    assert r2["compiled"] == 0 and r2["helper_processes"] == 0 and r2["disk_hits"] == r2["segments"] == r1["segments"]
    assert r2["code_bytes"] == r1["code_bytes"]
    # the same plan compiled in ONE process (jit_procs = 1): same file names (same keys), same code objects byte for byte
    r3 = run(n, mode, 1, {"QIP_HIP_CACHE_DIR": b})
    assert r3["compiled"] == r3["segments"] and r3["compiled_by_helpers"] == 0 and r3["helper_processes"] == 0
    assert files(a) == files(b)
    for f in files(a):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    # a damaged entry is a miss, not an error: it is compiled again and replaced
    victim = os.path.join(a, files(a)[0])
    good = open(victim, "rb").read()
    open(victim, "wb").write(good[: len(good) // 2])
    r4 = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": a})
    assert r4["compiled"] == 1 and r4["disk_hits"] == r4["segments"] - 1
    assert open(victim, "rb").read() == good


def test_no_cache_directory_and_no_helper_still_compile(tmp_path):
    n, mode = 14, 1 | 64
    r = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": "off"})
    assert r["dir"] == "" and r["disk_cache"] == 0 and r["compiled"] == r["segments"] and r["disk_stores"] == 0 and r["helper_processes"] == 0
    d = str(tmp_path / "c")
    r = run(n, mode, 4, {"QIP_HIP_CACHE_DIR": d, "QIP_HIP_JITC": "/nonexistent/qip_jitc"})
    assert r["compiled"] == r["segments"] and r["compiled_by_helpers"] == 0 and r["disk_stores"] == r["segments"]
    # a helper that dies: whatever it did not deliver is compiled in the calling process
    r = run(n + 1, mode, 4, {"QIP_HIP_CACHE_DIR": d, "QIP_HIP_JITC": "/bin/false"})
    assert r["compiled"] == r["segments"] and r["compiled_by_helpers"] == 0 and r["helper_processes"] >= 1


def test_compile_file_reports_errors(tmp_path):
    import rustqip_amd  # noqa: F401
    from rustqip_amd import _ffi

    bad = tmp_path / "bad.hip"
    bad.write_text('#include "qip_kernels.h"\nextern "C" __global__ void qip_segment() { this is not HIP; }\n')
    rc = _ffi.lib.qip_hip_jit_compile_file(str(bad).encode(), 0, str(tmp_path / "bad.co").encode())
    assert rc != 0 and "hiprtc could not compile" in _ffi.last_error()
    assert not os.path.exists(tmp_path / "bad.co")
    assert _ffi.lib.qip_hip_jit_compile_file(str(tmp_path / "missing.hip").encode(), 0, str(tmp_path / "x.co").encode()) != 0
    helper = os.path.join(ROOT, "rustqip_amd", "lib", "qip_jitc")
    p = subprocess.run([helper, "0", str(bad), str(tmp_path / "bad.co")], capture_output=True, text=True)
    assert p.returncode == 1 and "qip_jitc:" in p.stderr
    assert subprocess.run([helper], capture_output=True).returncode == 64
