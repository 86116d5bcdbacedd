import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def cbox_path():
    return os.path.join(ROOT, "scenes", "cbox", "scene.json")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle.lib()


@pytest.fixture(scope="session")
def hip_lib():
    """libakari_hip.so, built in-tree with hipcc if missing or stale (cross-compiles without a GPU)."""
    from akari_render_amd import build, capi

    build.build()
    return capi.lib()


@pytest.fixture(scope="session")
def ctx(hip_lib):
    """A real device context. GPU tests fail loudly (never skip, never fall back) if there is no GPU."""
    from akari_render_amd import capi

    return capi.Context(0)
