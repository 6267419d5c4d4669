import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hip_engine_factory():
    """Engines are cached per K for the whole session (one context per K)."""
    import bpmf_amd
    cache = {}

    def make(K, dtype="f64"):
        if (K, dtype) not in cache:
            cache[(K, dtype)] = bpmf_amd.HipEngine(K, dtype=dtype)
        return cache[(K, dtype)]
    yield make
    for e in cache.values():
        e.close()
