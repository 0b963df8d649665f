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


@pytest.fixture(params=[1, 0], ids=["lds_tree", "l1_nodes"])
def node_path(request, ctx):
    """Both node paths of the FAITHFUL render kernel: the LDS-resident-tree variant (one workgroup of 1024 threads per
    CU, the default for scenes whose node array fits a CU's LDS) and the 256-thread kernel that reads the node
    records through the vector L1 (tunable lds_tree = 0; what larger scenes get)."""
    default = ctx.get_tunable("lds_tree")
    ctx.set_tunable("lds_tree", request.param)  # 1: every tree that fits, however small
    yield request.param
    ctx.set_tunable("lds_tree", default)
