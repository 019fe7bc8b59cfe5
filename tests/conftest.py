import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """Build (if stale) and return the path of libpairnet_hip.so."""
    from pairnet_amd.build import build_lib
    return build_lib()
