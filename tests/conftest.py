import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP kernels through the C ABI)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def mc():
    """The product binding (ctypes over libmcull.so)."""
    return importlib.import_module("vulkan-renderer_amd")


@pytest.fixture(scope="session")
def H():
    return importlib.import_module("vulkan-renderer_amd.host")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_scene(name, oracle):
    z = np.load(os.path.join(GOLDEN, name + "_scene.npz"))
    return dict(vertices=z["vertices"].view(oracle.VERTEX), meshlet_data=z["meshlet_data"],
                meshlets=z["meshlets"].view(oracle.MESHLET))


@pytest.fixture(scope="session")
def kitten(O):
    return load_scene("kitten", O)


@pytest.fixture(scope="session")
def bunny(O):
    return load_scene("bunny", O)


@pytest.fixture(scope="session")
def config1(O):
    z = np.load(os.path.join(GOLDEN, "config1_inputs.npz"))
    return dict(globals=z["globals"].view(O.GLOBALS), transforms=z["transforms"].view(O.TRANSFORM))


def loaded_hip_runtime():
    """The HIP runtime this process has already loaded (torch's or the system's: whichever libmcull.so bound to), as a
    ctypes handle - not a second copy opened by soname."""
    import ctypes

    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


class Forced:
    """The library's evaluation switches (MCULL_* environment variables, read when a context is created) that the whole
    test run was started under.  The parity tests run under whatever is forced and compare outputs only; assertions on
    mechanism statistics (how many groups a level looked at, whether a level backed off) ask here whether the run has
    taken that mechanism away, and then expect what the switch leaves or skip."""

    def __init__(self):
        self.env = {k: v for k, v in os.environ.items() if k.startswith("MCULL_")}

    def off(self, name):
        return self.env.get(name, "")[:1] == "0"

    def on(self, name):
        return self.env.get(name, "")[:1] == "1"

    @property
    def streaming(self):  # the kernels for bounds larger than the caches: no verdicts handed on, hence no group level
        return self.on("MCULL_FORCE_STREAMING")

    @property
    def group_level(self):
        return not (self.off("MCULL_PRECULL_GROUPS") or self.off("MCULL_PRECULL_FLAGS") or self.streaming)

    @property
    def adaptive(self):
        return not self.off("MCULL_PRECULL_ADAPTIVE")


FORCED = Forced()  # at import: before any test changes the environment for a context of its own


@pytest.fixture(scope="session")
def forced():
    return FORCED


@pytest.fixture(scope="session")
def _shared_ctx(mc):
    # the instance pre-cull from 2048 groups on (the library's own threshold is the size from which it pays):
    # the scenes of the parity tests then run with it, as the large scenes do
    old = os.environ.get("MCULL_PRECULL_MIN_GROUPS")
    os.environ["MCULL_PRECULL_MIN_GROUPS"] = "2048"
    try:
        c = mc.Context(0)
    finally:
        if old is None:
            del os.environ["MCULL_PRECULL_MIN_GROUPS"]
        else:
            os.environ["MCULL_PRECULL_MIN_GROUPS"] = old
    yield c
    c.close()


@pytest.fixture()
def ctx(_shared_ctx):
    """The session's context (creating one per test would dominate the run time), handed out per test and put
    back into a known state afterwards whatever the test did or wherever it failed: no instance ranges, no
    timing, view slot 0 selected by the next cull.  Uploads are each test's own business (every test uploads
    what it culls)."""
    yield _shared_ctx
    try:
        _shared_ctx.set_instance_ranges(None, None)
        _shared_ctx.enable_timing(False)
        _shared_ctx.synchronize()
    except Exception:  # a context broken by the test fails the next test's own calls, with its own message
        pass
