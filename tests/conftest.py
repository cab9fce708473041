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
    return smm_jl_amd
