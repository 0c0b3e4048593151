import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a full-size (n >= 26) or multi-process test; collected LAST, so that a run cut "
                                       "short by a time limit has already been through one cheaper test of every SURVEY §8 row")


def _have_gpu() -> bool:
    try:
        import rustqip_amd

        return rustqip_amd.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no CPU fallback.
    # (Tests marked gpu are simply deselected by `-m "not gpu"` on the CPU container.)
    # Order: the cheap tests — at least one per SURVEY §8 row — first, the full-size and multi-process ones
    # (marker `slow`) last, in file order within each group: a kill at a time limit then costs depth, not rows.
    items.sort(key=lambda it: 1 if it.get_closest_marker("slow") else 0)  # (stable)


@pytest.fixture(scope="session")
def oracle():
    from oracle import qip_oracle

    return qip_oracle


def has_tuning_options() -> bool:
    """True when libqip_hip.so was built with -DQIP_HIP_TUNING (QIP_HIP_TUNING=1 python -m rustqip_amd.build): the measured
    alternatives of earlier rounds are switchable again.  The product build fixes them at their defaults; the A/B tests of
    those alternatives skip there (`needs_tuning`)."""
    import ctypes

    from rustqip_amd import _ffi

    return _ffi.lib.qip_hip_set_global_option(b"tile_diag_runs", ctypes.c_int64(1)) == 0


def needs_tuning():
    if not has_tuning_options():
        pytest.skip("an A/B test of a measured alternative: needs a -DQIP_HIP_TUNING build of libqip_hip.so")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (the checker), under the name the GPU parity tests use"""
    from oracle import qip_oracle

    return qip_oracle
