"""Run-time specialised kernels (oh_specialize*, optas_amd/csrc/oh_jit.hip): the kernels that walk the kinematic chain in their inner loops
are compiled once more with hiprtc behind a constexpr copy of the handle's chain.  The generic kernels stay the reference path: both must
produce the same iterates (same header text; folded constants only drop multiplications by exact zeros and ones)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import KUKA_KIN, MED7_KIN, SEED, oh_debug
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import KinematicsHandle, RobotModel

LINK = "end_effector_ball"
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])


def test_compiles_without_a_device_and_fills_the_cache(tmp_path, monkeypatch):
    """hiprtc is a compiler: like hipcc it needs no GPU.  The code object lands in $OPTAS_HIP_CACHE and is found there the second time."""
    monkeypatch.setenv("OPTAS_HIP_CACHE", str(tmp_path / "cache"))
    chain = RobotModel(urdf_filename=MED7_KIN).kinematic_chain("lbr_link_ee")
    a = _lib.specialize_compile(chain)
    files = glob.glob(str(tmp_path / "cache" / "spec_*.hsaco"))
    assert not a["from_disk_cache"] and len(files) == 2  # solver kernels + K1
    assert all(open(f, "rb").read(4) == b"\x7fELF" for f in files)
    b = _lib.specialize_compile(chain)
    assert b["from_disk_cache"] and b["seconds"] < a["seconds"]
    # another chain is another code object
    _lib.specialize_compile(RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK))
    assert len(glob.glob(str(tmp_path / "cache" / "spec_*.hsaco"))) == 4


def test_compile_rejects_chains_the_solver_kernels_do_not_take():
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain("lwr_arm_3_link")  # 3 of 7 joints: K1 only, no solver kernels
    lib = _lib.load()
    assert lib.oh_specialize_compile(C.byref(chain), None) == _lib.OH_ERR_INVALID and b"every model joint" in lib.oh_last_error()
    assert lib.oh_specialize_compile(None, None) == _lib.OH_ERR_INVALID
    assert lib.oh_specialize(None) == _lib.OH_ERR_INVALID


def _backend(T=50, **kw):
    from oracle.problems import FigureEightNLP
    from oracle.robot import OracleRobot

    nlp = FigureEightNLP(OracleRobot(KUKA_KIN), LINK, T=T)
    robot = RobotModel(urdf_filename=KUKA_KIN)
    return nlp, FigureEightBackend(robot.kinematic_chain(LINK), T, nlp.dt, nlp.local_path.T, max_iter=300, tol=1e-6, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 48, 6000])  # tail kernel alone; tail for a small batch; batched kernels + compaction + hand-over to the tail
def test_specialised_solver_kernels_match_the_generic_ones(hip_lib, monkeypatch, B):
    oh_debug(monkeypatch, specialize="0")
    oh_debug(monkeypatch, tail_threshold="2048")  # B = 6000 through the batched kernels (the default hands it to the tail kernel as a whole)
    nlp, gen = _backend()
    rng = np.random.default_rng(SEED + 31)
    qc = QC0 + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.1, 0.1, (B - 1, 7))])
    x0 = np.stack([nlp.seed(q) for q in qc[:64]])
    x0 = x0[np.arange(B) % len(x0)].copy()
    x0[:, : 7 * 50] = np.tile(qc, (1, 50))
    rg = gen.solve(x0, qc)
    assert not gen.specialize_info()["loaded"]
    _, spe = _backend()
    info = spe.specialize()
    assert info["loaded"] and info["fk_jac_loaded"]
    rs = spe.solve(x0, qc)
    assert (rg.status == 0).all() and (rs.status == 0).all()
    # same text, same operation order: the iterates agree to rounding, so do the step counts except where a ratio test is borderline
    # (a long run that parts at one borderline step may end a few steps apart, like the GPU against the host port)
    assert (rg.iters == rs.iters).mean() >= 0.99 and (np.abs(rg.iters.astype(int) - rs.iters) <= np.maximum(3, rg.iters // 4)).all()
    same = rg.iters == rs.iters
    assert np.abs(rg.f[same] - rs.f[same]).max() <= 1e-11 * np.abs(rg.f).max()
    dx = np.abs(rg.x[same] - rs.x[same]).max(1)
    # stopping at |Z^T G| <= 1e-6 leaves ~1e-5 rad of play along the weakly curved elbow-swivel direction for the few that parted on the way
    assert np.quantile(dx, 0.99) <= 1e-9 and dx.max() < 1e-4
    assert np.abs(rg.f - rs.f).max() <= 1e-8 * np.abs(rg.f).max()
    # the handle reports the code objects it launches
    ks, kg = spe.kernel_info("k_evalb"), _lib.kernel_info("k_evalb")
    assert ks["registers_per_lane"] > 0 and ks["block"] == kg["block"] == 256
    gen.close()
    spe.close()


@pytest.mark.gpu
def test_automatic_specialisation_threshold(hip_lib, monkeypatch, tmp_path):
    """Automatic mode: kernels are compiled for the chain at the first batch of >= 4096 instances (seconds of hiprtc) -- or at the first solve of
    any size when the code object is already in the disk cache (milliseconds: a controller that only ever solves one instance gets them too)."""
    oh_debug(monkeypatch, specialize=None)
    T = 46  # (a horizon of its own: the process-wide map of loaded code objects is keyed by the chain, not by T, so use a fresh cache AND check it)
    monkeypatch.setenv("OPTAS_HIP_CACHE", str(tmp_path / "cache"))
    chain = RobotModel(urdf_filename=MED7_KIN).kinematic_chain("lbr_link_ee")  # not loaded by any other test of this process
    from oracle.problems import FigureEightNLP
    from oracle.robot import OracleRobot

    nlp = FigureEightNLP(OracleRobot(MED7_KIN), "lbr_link_ee", T=T)
    qc0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    rng = np.random.default_rng(SEED + 32)

    def run(be, B):
        qc = qc0 + rng.uniform(-0.05, 0.05, (B, 7))
        x0 = np.zeros((B, nlp.nx))
        x0[:, : 7 * T] = np.tile(qc, (1, T))
        r = be.solve(x0, qc)
        assert (r.status == 0).mean() > 0.99
        return be.specialize_info()["loaded"]

    be = FigureEightBackend(chain, T, nlp.dt, nlp.local_path.T, max_iter=300, tol=1e-6)
    assert [run(be, B) for B in (64, 4096, 8)] == [False, True, True]  # loaded at the first batch of >= 4096 instances, kept afterwards
    be.close()
    assert len(glob.glob(str(tmp_path / "cache" / "spec_*.hsaco"))) >= 1
    be2 = FigureEightBackend(chain, T, nlp.dt, nlp.local_path.T, max_iter=300, tol=1e-6)
    assert run(be2, 1)  # the object is at hand: loaded at the first solve, whatever its size
    be2.close()


@pytest.mark.gpu
def test_specialised_fk_jac_matches_the_generic_kernel(hip_lib, monkeypatch):
    rng = np.random.default_rng(SEED + 33)
    for kin, link in ((KUKA_KIN, LINK), (KUKA_KIN, "lwr_arm_3_link"), (MED7_KIN, "lbr_link_ee")):
        chain = RobotModel(urdf_filename=kin).kinematic_chain(link)
        Q = rng.uniform(-2.0, 2.0, (5000, chain.ndof))
        oh_debug(monkeypatch, specialize="0")
        pg, Jg = KinematicsHandle(chain).fk_jac(Q)
        oh_debug(monkeypatch, specialize="1")
        h = KinematicsHandle(chain)
        ps, Js = h.fk_jac(Q)
        info = (C.c_double * 4)()
        assert hip_lib.oh_specialize_info(h._h, info) == 0 and info[1] == 1.0
        assert np.abs(pg - ps).max() <= 1e-15 and np.abs(Jg - Js).max() <= 1e-15


@pytest.mark.gpu
def test_specialize_errors(hip_lib):
    from optas_amd.backend import PointMassBackend

    pm = PointMassBackend()
    assert hip_lib.oh_specialize(pm._h) == _lib.OH_ERR_STATE and b"oh_set_constants" in hip_lib.oh_last_error()
    pm.close()
