import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer statistical run")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def oracle_engine(monkeypatch):
    """CPU tests of the host-side classes: the oracle engine (tests/_oracle_engine.py) replaces the HIP engine behind
    bipymc_amd.demc / samplers by patching the module attribute -- the sampler classes take no engine argument."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import _oracle_engine
    import bipymc_amd.demc as D
    monkeypatch.setattr(D, "_engine_factory", _oracle_engine.factory)
    return _oracle_engine.factory
