import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (MHC_4 end-to-end through the oracle)")


def _make(path, *targets):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, path), *targets])


def pytest_sessionstart(session):
    """a fresh checkout has no built artefacts, and test modules import dipgenie_amd.capi (which refuses to load without
    libdipgenie_hip.so) at collection time: cross-compile the HIP library first (hipcc needs no GPU)"""
    if not os.path.exists(os.path.join(ROOT, "dipgenie_amd", "csrc", "libdipgenie_hip.so")):
        _make("dipgenie_amd/csrc")


@pytest.fixture(scope="session")
def built_cpu():
    """oracle restatement, host pipeline objects and the host+oracle harness (CPU only)."""
    _make("oracle", "restate")
    _make("dipgenie_amd/host", "host_only")
    _make("tests/harness")
    return os.path.join(ROOT, "tests", "harness", "dg_host_oracle")


@pytest.fixture(scope="session")
def built_hip():
    """libdipgenie_hip.so + product CLI (hipcc cross-compiles without a GPU)."""
    _make("dipgenie_amd/csrc")
    _make("dipgenie_amd/host")
    return os.path.join(ROOT, "bin", "DipGenie")


@pytest.fixture(scope="session")
def gpu_ctx(built_hip):
    from dipgenie_amd import capi
    ctx = capi.Context(0)   # raises loudly without a gfx950 device: no fallback
    yield ctx
    ctx.close()
