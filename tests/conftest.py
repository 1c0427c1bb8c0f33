import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (checker)."""
    from oracle import pyoracle
    return pyoracle.load()


@pytest.fixture(scope="session")
def vdb():
    """The product library (HIP).  Only GPU-marked tests may call compute entry points."""
    import vectordb_amd
    return vectordb_amd.api()


def _backend(request, name):
    if name == "oracle":
        return request.getfixturevalue("ora")
    return request.getfixturevalue("vdb")


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    """Backend under test: the oracle on CPU, the HIP library on the GPU box."""
    api = _backend(request, request.param)
    api.backend = request.param
    return api


@pytest.fixture(params=["oracle", "hip"])
def hostbe(request):
    """Host-only logic (payloads, filters, inverted index): no GPU compute behind these calls."""
    api = _backend(request, request.param)
    api.backend = request.param
    return api
