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


def pytest_collection_modifyitems(config, items):
    """GPU runs: bring PyTorch's HIP context up BEFORE the first kernel of this library runs.  Some tests (the sharded
    filter) import torch only late in the session; twice on the GPU pool the first `import torch` + stream creation after
    tens of tests' worth of HIP work in the same process stalled for minutes, while the order "torch first" (what bench.py
    and every isolated test run do) never has."""
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda").item()
    except Exception:   # no torch / no GPU: the tests that need them say so themselves
        pass
