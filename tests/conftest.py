"""pytest configuration: markers and shared fixture loaders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def load_golden(name: str):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def g_mini_small():
    return load_golden("minibatch_small.npz")


@pytest.fixture(scope="session")
def g_mini_dense():
    return load_golden("minibatch_dense.npz")


@pytest.fixture(scope="session")
def g_full_reddit():
    return load_golden("fullgraph_reddit_like.npz")


@pytest.fixture(scope="session")
def g_full_amazon():
    return load_golden("fullgraph_amazon_like.npz")
