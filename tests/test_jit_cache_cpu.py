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
if os.environ.get("QIP_TEST_IMPORT_TORCH_FIRST"):
    import torch  # brings PyTorch's own bundled libhiprtc / libamd_comgr / HIP runtime into the process
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


def test_a_host_program_with_its_own_rocm_libraries_shares_the_cache(tmp_path):
    """r6: bench.py imports torch, whose wheel carries another ROCm release's libhiprtc / comgr / runtime.  The key used to name the
    compiler "as loaded in this process": the helpers' (system ROCm) objects then failed the requester's header check and every
    segment was compiled twice (compiled_by_helpers = 0 in the driver's bench), and a process without torch never found what one
    with torch had cached.  The key names the installation the helpers use: same file names either way, helpers' work accepted,
    hits across the two kinds of process; a single new segment of such a process goes to a helper too."""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    n, mode = 16, 1 | 64
    with_torch = {"QIP_TEST_IMPORT_TORCH_FIRST": "1"}
    r1 = run(n, mode, -1, dict(with_torch, QIP_HIP_CACHE_DIR=a))
    assert r1["compiled"] == r1["segments"] >= 3 and r1["disk_hits"] == 0
    if r1["procs"] > 1:
        assert r1["compiled_by_helpers"] == r1["segments"], r1
    r2 = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": b})
    assert files(a) == files(b)
    for f in files(a):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    r3 = run(n, mode, -1, {"QIP_HIP_CACHE_DIR": a})  # no torch, reads what the torch process left
    assert r3["compiled"] == 0 and r3["disk_hits"] == r3["segments"] == r1["segments"]
    r4 = run(n, mode, -1, dict(with_torch, QIP_HIP_CACHE_DIR=b))  # and the other way round
    assert r4["compiled"] == 0 and r4["disk_hits"] == r4["segments"]
    # one damaged entry = ONE new segment: in a process that has a foreign libhiprtc loaded it is still a helper that compiles it
    victim = os.path.join(a, files(a)[0])
    good = open(victim, "rb").read()
    open(victim, "wb").write(good[:100])
    r5 = run(n, mode, -1, dict(with_torch, QIP_HIP_CACHE_DIR=a))
    assert r5["compiled"] == 1 and r5["compiled_by_helpers"] == 1 and r5["helper_processes"] == 1, r5
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


def test_cache_trusts_only_what_this_user_wrote(tmp_path):
    """r6: a code object found on disk ends up on the GPU, so anything odd about it is a MISS and a
    recompilation, never a load: a flipped byte in the code (the header's hash of the CODE bytes), a truncated-then-padded file,
    a planted file with a valid-looking header, a symlink, a group/world-writable file or directory."""
    d = str(tmp_path / "c")
    n, mode = 15, 1 | 64
    r1 = run(n, mode, 1, {"QIP_HIP_CACHE_DIR": d})
    segs = r1["segments"]
    assert r1["compiled"] == segs and len(files(d)) == segs and segs >= 3
    assert oct(os.stat(d).st_mode & 0o777) == "0o700" and all(oct(os.stat(os.path.join(d, f)).st_mode & 0o777) == "0o600" for f in files(d))
    names = files(d)
    good = {f: open(os.path.join(d, f), "rb").read() for f in names}

    def second(expect_compiled, expect_hits):
        r = run(n, mode, 1, {"QIP_HIP_CACHE_DIR": d})
        assert (r["compiled"], r["disk_hits"]) == (expect_compiled, expect_hits), r
        for f in names:  # whatever was damaged has been replaced by a good object
            assert open(os.path.join(d, f), "rb").read() == good[f], f
            assert not os.path.islink(os.path.join(d, f))

    victim = os.path.join(d, names[0])
    # one flipped byte in the middle of the code: same length, same header
    b = bytearray(good[names[0]])
    b[len(b) // 2] ^= 0x40
    open(victim, "wb").write(bytes(b))
    second(1, segs - 1)
    # truncated, then padded back to its length with zeros
    open(victim, "wb").write(good[names[0]][: len(b) // 2] + bytes(len(b) - len(b) // 2))
    second(1, segs - 1)
    # another entry's (valid) object planted under this name: the key's second word in the header does not match
    open(victim, "wb").write(good[names[1]])
    second(1, segs - 1)
    # a symlink to a valid object is not followed
    os.unlink(victim)
    os.symlink(os.path.join(d, names[1]), victim)
    second(1, segs - 1)
    # a file others may write is not trusted
    os.chmod(victim, 0o666)
    second(1, segs - 1)
    os.chmod(victim, 0o600)
    second(0, segs)
    # a directory others may write is not used at all (no loads, no stores); explicit selection reports why
    os.chmod(d, 0o777)
    r = run(n, mode, 1, {"QIP_HIP_CACHE_DIR": d})
    assert r["dir"] == "" and r["disk_cache"] == 0 and r["compiled"] == segs and r["disk_hits"] == 0 and r["disk_stores"] == 0
    import rustqip_amd  # noqa: F401
    from rustqip_amd import _ffi

    assert _ffi.lib.qip_hip_jit_set_cache_dir(d.encode()) != 0 and "writable by group or others" in _ffi.last_error()
    os.chmod(d, 0o700)
    assert _ffi.lib.qip_hip_jit_set_cache_dir(d.encode()) == 0
    assert _ffi.lib.qip_hip_jit_set_cache_dir(None) == 0


def test_cache_directory_is_bounded(tmp_path):
    """r6: after a store the directory is trimmed to its bound, oldest modification time first; a hit refreshes the time"""
    d = str(tmp_path / "c")
    mode = 1 | 64
    r1 = run(15, mode, 1, {"QIP_HIP_CACHE_DIR": d})
    first = files(d)
    total = sum(os.path.getsize(os.path.join(d, f)) for f in first)
    assert r1["disk_trimmed"] == 0 and total > 0
    old = 1_000_000_000
    for i, f in enumerate(first):  # age them, the first file least
        os.utime(os.path.join(d, f), (old + 100 - i, old + 100 - i))
    # a second plan under a bound of 1 MiB: the new objects stay, the oldest of the first plan go
    r2 = run(16, mode, 1, {"QIP_HIP_CACHE_DIR": d, "QIP_HIP_CACHE_MAX_MB": "1"})
    now = files(d)
    kept_total = sum(os.path.getsize(os.path.join(d, f)) for f in now)
    if total + r2["compiled"] * 1 > 0 and r2["disk_trimmed"]:
        assert kept_total <= (1 << 20)
        gone = [f for f in first if f not in now]
        assert gone and gone == first[len(first) - len(gone):]  # the oldest (highest index: aged most) went first
    else:  # (the two plans together fit one MiB on this compiler: nothing to trim)
        assert kept_total <= (1 << 20)
    # stale temporaries of a killed process are swept with a store; fresh ones are left alone
    stale, fresh = os.path.join(d, "seg.1.2.3.hip"), os.path.join(d, "seg.4.5.6.hip")
    open(stale, "w").write("x")
    open(fresh, "w").write("x")
    os.utime(stale, (old, old))
    run(17, mode, 1, {"QIP_HIP_CACHE_DIR": d})
    assert not os.path.exists(stale) and os.path.exists(fresh)


def test_helper_removes_its_sources_when_asked(tmp_path):
    """the background jobs of one-shot callers (option tile_auto): qip_jitc -u leaves the code object and removes the source"""
    import rustqip_amd  # noqa: F401
    from rustqip_amd import circuits
    from rustqip_amd.ops import debug_tile_jit

    src = debug_tile_jit(14, circuits.h_layer(14) + circuits.c2_random_circuit(14, 40, seed=3), 1 | 64)["first_source"]
    f = tmp_path / "seg.hip"
    f.write_text(src)
    helper = os.path.join(ROOT, "rustqip_amd", "lib", "qip_jitc")
    p = subprocess.run([helper, "-u", "0", str(f), str(tmp_path / "seg.co")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert not f.exists() and (tmp_path / "seg.co").stat().st_size > 1000
    assert subprocess.run([helper, "-u"], capture_output=True).returncode == 64
