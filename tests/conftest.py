import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (test infrastructure; never loaded by the product)."""
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def hiplib():
    """The product library through its C ABI (ctypes)."""
    from modppl_amd import capi

    return capi.load()
