import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU replay oracle (test infrastructure), built on demand with g++."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name))   # allow_pickle=False (default)
    return load
