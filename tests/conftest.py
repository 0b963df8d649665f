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


@pytest.fixture(params=["wavefront", "hybrid", "lds_tree", "l1_nodes"])
def node_path(request, ctx):
    """The forms of the FAITHFUL render kernel: the path-pool kernel (srt_wavefront.hip: lanes traverse the
    LDS-resident threaded tree, full waves shade contexts taken from per-class rings; the default for scenes whose node
    array fits a CU's LDS), its hybrid form (only the tree's top in LDS, the other records read from global memory: what
    trees too large for LDS get; here forced onto every tree of more than 24 nodes with the tunable wf_resident_max), the
    step-scheduler kernel over the LDS-resident tree (tunable wavefront = 0), and the 256-thread step-scheduler kernel
    that reads the node records through the vector L1 (lds_tree = 0).  A counting launch (count_stats) of the first two
    runs the third: the counters belong to the scene."""
    saved = {k: ctx.get_tunable(k) for k in ("lds_tree", "wavefront", "wf_resident_max")}
    ctx.set_tunable("lds_tree", 0 if request.param == "l1_nodes" else 1)  # 1: every tree that fits, however small
    ctx.set_tunable("wavefront", 1 if request.param in ("wavefront", "hybrid") else 0)
    ctx.set_tunable("wf_resident_max", 24 if request.param == "hybrid" else 0)  # read at upload: the tests upload after this
    yield request.param
    for k, v in saved.items():
        ctx.set_tunable(k, v)


def render_counted(ctx, p, node_path):
    """ctx.render_image(p) for a test that also reads ctx.stats(): the image comes from the kernel under test, the
    counters from the counting variant.  For the path-pool kernel these are two launches (its counting variant is the
    step-scheduler kernel over the same tree), and the image is checked to come from the path-pool kernel itself."""
    if node_path not in ("wavefront", "hybrid") or not p.countStats:
        return ctx.render_image(p)
    p.countStats = 0
    acc, rgba = ctx.render_image(p)
    info = ctx.launch_info()
    assert info["wavefront"] or not info["lds_tree"], info  # a world without a tree (or with one that does not fit) has no LDS-resident form
    p.countStats = 1
    ctx.render_image(p)
    return acc, rgba
