import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the checkers (oracle/) and the product library once per session.  The product library is
    cross-compiled for gfx950 by hipcc, which works without a GPU."""
    import __graft_entry__ as ge

    ge.build()
    return True


@pytest.fixture(scope="session")
def port(built):
    from oracle.pyoracle import Oracle

    return Oracle("port")


@pytest.fixture(scope="session")
def reference(built):
    from oracle.pyoracle import Oracle, available

    if not available("reference"):
        pytest.skip("oracle/_ref/libtutu_ref.so not built (needs /root/reference)")
    return Oracle("reference")


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
