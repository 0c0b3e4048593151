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
