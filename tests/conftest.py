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


@pytest.fixture
def diag(monkeypatch):
    """Tests that drive an A/B or test switch of the library (MP_K1_MT, MP_DEFERRED_LOOKUPS, MP_FUSED_DRAWS, ...) run against its
    DIAGNOSTICS build (libmodppl_hip_diag.so: the same sources compiled with -DMP_DIAGNOSTICS, csrc/mp_diag.h): the product library
    reads no environment variable, so a switch set in front of it would select nothing.  Every handle the test creates belongs to
    that build; the product library stays loaded and is what every other test runs."""
    from modppl_amd import capi

    monkeypatch.setattr(capi, "_lib", capi.load_diag())
    return monkeypatch


def diag_env(env):
    """for tests that run workers in subprocesses: the same choice there, when the environment handed to the workers sets a switch the
    library itself reads (the workers' `modppl_amd.capi.load()` honours MODPPL_HIP_LIB)"""
    from modppl_amd import build as B
    from modppl_amd import capi

    if any(k in env for k in ("MP_SHARD_OWNED_CAP", "MP_SHARD_OWNED_FIXED_MAX_BYTES", "MP_SHARD_FIXED", "MP_SHARD_SELF", "MP_K1_MT", "MP_FUSED_DRAWS")):
        capi.load_diag()   # (built if stale)
        env["MODPPL_HIP_LIB"] = B.SO_DIAG
    return env


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
