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


_OPTION_ALIASES = {"tq_jac": "tq_jac_dual"}
_WORDS = {"dual": 1.0, "global": 0.0, "lds": 1.0, "auto": 2.0}


def oh_debug(monkeypatch, **kw):
    """Options for the handles created from here on, through the library's one environment hook (OH_DEBUG_OPTIONS = "name=value,..."; per-handle
    oh_set_option is what product code uses).  Names as in include/optas_hip.h (an OH_ prefix / upper case is accepted); value None removes a name."""
    cur = {}
    for item in filter(None, os.environ.get("OH_DEBUG_OPTIONS", "").split(",")):
        k, v = item.split("=")
        cur[k] = v
    for k, v in kw.items():
        k = k.lower()
        k = k[3:] if k.startswith("oh_") else k
        k = _OPTION_ALIASES.get(k, k)
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = repr(float(_WORDS.get(v, v) if isinstance(v, str) else v))
    if cur:
        monkeypatch.setenv("OH_DEBUG_OPTIONS", ",".join(f"{k}={v}" for k, v in cur.items()))
    else:
        monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
