import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_rules():
    return load_golden("rules.npz")


@pytest.fixture(scope="session")
def golden_boards():
    return load_golden("boards_csv.npz")


@pytest.fixture(scope="session")
def golden_mcts():
    return load_golden("mcts.npz")


@pytest.fixture(scope="session")
def golden_nn():
    return load_golden("nn.npz")


@pytest.fixture(scope="session")
def golden_selfplay():
    return load_golden("selfplay.npz")
