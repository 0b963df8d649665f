import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def srt():
    return importlib.import_module("sexy-raytracer_amd")


@pytest.fixture(scope="session")
def abi(srt):
    return srt.abi


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Builds oracle/liboracle.so on first use."""
    import oracle.oracle_py as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def dev(srt):
    """The C-ABI library binding; importing fails loudly if the HIP extension is missing."""
    so = os.path.join(ROOT, "sexy-raytracer_amd", "csrc", "libsrt_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return srt.device()


@pytest.fixture(scope="session")
def ctx(dev):
    c = dev.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def camera(dev, abi):
    return dev.make_camera(abi.default_camera_params())
