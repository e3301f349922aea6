import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# code objects of the run-time specialised kernels (oh_specialize) stay inside the tree
os.environ.setdefault("OPTAS_HIP_CACHE", os.path.join(ROOT, ".optas_hip_cache"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
KUKA_KIN = os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json")
MED7_KIN = os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json")
TESTER_KIN = os.path.join(GOLDEN, "tester_robot.kin.json")
SEED = 20260927


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_fk():
    return np.load(os.path.join(GOLDEN, "fk_golden.npz"))


@pytest.fixture(scope="session")
def golden_nlp():
    return np.load(os.path.join(GOLDEN, "nlp_golden.npz"))


@pytest.fixture(scope="session")
def golden_sm():
    return np.load(os.path.join(GOLDEN, "spatialmath_golden.npz"))


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree liboptas_hip.so; GPU tests must run the native library, never a fallback."""
    from optas_amd import _lib

    lib = _lib.load()
    if _lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible: liboptas_hip has no CPU path")
    return lib
