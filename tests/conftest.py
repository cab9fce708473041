import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (test infrastructure)"""
    from oracle import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def S():
    import smm_jl_amd
    if os.environ.get("SMM_TEST_BUILD") == "hooks":   # the whole suite against the test build (no seam set: the same selection logic)
        smm_jl_amd._abi.use_test_hooks(True)
    return smm_jl_amd


@pytest.fixture
def hooks(monkeypatch):
    """contexts created inside the test come from the test build of the library (libsmmhip_hooks.so): the SMMHIP_* seams that
    force a kernel or switch a fast path off do not exist in the shipped libsmmhip.so"""
    from smm_jl_amd import _abi as A
    monkeypatch.setattr(A, "_lib", A.load_hooks())
    yield A
