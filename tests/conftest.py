import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def built():
    """Builds the product library and the oracle once per session (CPU-only operation)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import pyoracle
    pyoracle.lib()
    return pyoracle
