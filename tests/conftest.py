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


@pytest.fixture(scope="session")
def agents_golden():
    return np.load(os.path.join(GOLDEN, "agents_golden.npz"))


def golden_cases(npz, prefix):
    """Names of the recorded cases of one agent kind, e.g. prefix 'mcts_' -> ['d2_s3_graph', ...]."""
    return sorted(k[len(prefix):-len("_params")] for k in npz.files if k.startswith(prefix) and k.endswith("_params"))


@pytest.fixture(scope="session")
def standin_net(agents_golden):
    from standin_net import StandInNet
    return StandInNet(weights={k[4:]: agents_golden[k] for k in agents_golden.files
                               if k.startswith("net_") and not k.startswith("net_probe")})
