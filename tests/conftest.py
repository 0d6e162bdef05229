import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


# The diagnostics build of the library (-DCK_DIAG: chalkydri_amd/csrc/Makefile): the only one in which the path-forcing knobs
# (CK_FMERGE_CAP, CK_FIT_FLAT, CK_PARTS, ...) exist.  Tests that force a path run a child process against it; everything
# else in the suite runs against the product library, which ignores those variables.
DIAG_LIB = os.path.join(ROOT, "chalkydri_amd", "lib", "diag", "libchalkydri_hip.so")


def diag_env(**extra):
    env = dict(os.environ, CHALKYDRI_HIP_LIB=DIAG_LIB)
    env.update(extra)
    return env


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
