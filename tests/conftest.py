import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def chr22():
    from oracle import oracle as O
    return dict([O.pseudo_reference(os.path.join(GOLDEN, "pseudo_chr22.npz"))])


@pytest.fixture(scope="session")
def chr1():
    from oracle import oracle as O
    return dict([O.pseudo_reference(os.path.join(GOLDEN, "pseudo_chr1.npz"))])
