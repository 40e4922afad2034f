import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("slam-eds_amd.synth")


@pytest.fixture(scope="session")
def capi():
    mod = importlib.import_module("slam-eds_amd.capi")
    mod.build()          # hipcc cross-compiles for gfx950 without a GPU
    return mod


@pytest.fixture(scope="session")
def po():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def npo():
    import np_oracle
    return np_oracle


@pytest.fixture(scope="session")
def gpu(capi):
    """GPU tests fail loudly (never skip, never fall back) when no device is visible."""
    n = capi.device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests need an MI355X (libeds_hip has no CPU fallback)"
    return n
