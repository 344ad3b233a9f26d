"""pytest configuration: markers, paths, shared fixtures."""

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Build libdsx_hip.so in-tree if it is missing or older than any of its sources (hipcc cross-compiles
    without a GPU; on the GPU box the library travels with the snapshot and is up to date)."""
    import __graft_entry__ as g

    g.build()
    yield


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_small():
    return _load("small_full.npz")


@pytest.fixture(scope="session")
def golden_large():
    return _load("large_stats.npz")


@pytest.fixture(scope="session")
def golden_misc():
    return _load("misc.npz")


@pytest.fixture(scope="session")
def golden_sweep():
    return _load("sweep.npz")
