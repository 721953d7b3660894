import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rl-rubiks_amd")           # holds the drop-in `librubiks` package + csrc/
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "cube_golden.npz"))


@pytest.fixture(scope="session")
def bfs_golden():
    return np.load(os.path.join(GOLDEN, "bfs_golden.npz"))
