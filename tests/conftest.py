import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    return _orc.load()


@pytest.fixture(scope="session")
def golden_js():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "pathtracer_js_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def rt():
    """The product package (ctypes binding of libmi355pt.so)."""
    import importlib
    return importlib.import_module("raytracer-public_amd")


@pytest.fixture
def gpu_ctx(rt):
    """A fresh context per test: batch size, external buffers, tile shares and accumulation state are per context, so nothing a
    test sets (or leaves behind when it fails half way) reaches the next one."""
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
