import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hl():
    import halo2_lasso_amd
    return halo2_lasso_amd


@pytest.fixture(scope="session")
def ctx(hl):
    """Device context; fails loudly (no fallback) when there is no GPU."""
    c = hl.Context(0)
    yield c
    c.close()
