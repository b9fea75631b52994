import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rta():
    """the product package (directory name has a hyphen, so it is imported by name)"""
    return importlib.import_module("ray-tracing_amd")


@pytest.fixture(scope="session")
def scenes(rta):
    return importlib.import_module("ray-tracing_amd.scenes")


@pytest.fixture(scope="session")
def ob():
    """oracle bindings (CPU restatement + real reference where built)"""
    from oracle import bindings
    return bindings


@pytest.fixture(scope="session")
def gpu(rta):
    r = rta.Renderer(0)
    yield r
    r.close()
