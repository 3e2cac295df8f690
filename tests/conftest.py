import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    """A ROCm device node is present (no torch import: that costs a minute on a cold box)."""
    import glob
    return os.path.exists("/dev/kfd") and bool(glob.glob("/dev/dri/renderD*"))


def pytest_collection_modifyitems(config, items):
    if _gpu_visible():
        return
    # `-m gpu` on a box without a GPU must not come out green with nothing run (MM_ALLOW_NO_GPU=1: skip them knowingly)
    expr = (config.getoption("-m") or "").strip()
    if "gpu" in expr and "not gpu" not in expr and os.environ.get("MM_ALLOW_NO_GPU") != "1":
        raise pytest.UsageError("-m gpu was asked for and no MI355X is visible (/dev/kfd or /dev/dri/renderD* missing); "
                                "set MM_ALLOW_NO_GPU=1 to skip the GPU tests knowingly")
    skip = pytest.mark.skip(reason="no MI355X visible (/dev/kfd missing): GPU parity tests need the HIP path, which has no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def chr22():
    from oracle import oracle as O
    return dict([O.pseudo_reference(os.path.join(GOLDEN, "pseudo_chr22.npz"))])


@pytest.fixture(scope="session")
def chr1():
    from oracle import oracle as O
    return dict([O.pseudo_reference(os.path.join(GOLDEN, "pseudo_chr1.npz"))])
