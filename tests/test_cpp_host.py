"""The C++ host mirror (rustqip_amd/host/qip_hip.hpp) — compiled with g++ against libqip_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    from oracle import qip_oracle  # noqa: F401  (makes sure libqip_oracle.so exists for the link)

    subprocess.run(["make", "-C", CPP, "-B", "test_host_mirror"], check=True, capture_output=True)
    return os.path.join(CPP, "test_host_mirror")


def test_cpp_host_mirror_host_logic():
    exe = _build()
    res = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "PASSED" in res.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_gpu():
    exe = _build()
    res = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "PASSED" in res.stdout and "max|delta| vs oracle" in res.stdout
