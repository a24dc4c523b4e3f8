import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    from oracle import loader

    return loader.port()


@pytest.fixture(scope="session")
def ref():
    from oracle import loader

    lib = loader.ref()
    if lib is None:
        pytest.skip("oracle/_ref (the real reference) is not built on this machine")
    return lib


@pytest.fixture(scope="session")
def hip_ctx():
    """One HIP context for the whole GPU session (fails loudly if the .so or the GPU is missing)."""
    from libultrahdr_amd.ultrahdr import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
