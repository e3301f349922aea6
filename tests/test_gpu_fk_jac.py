"""K1 (oh_fk_jac*) through the C ABI against the literal oracle and the committed golden vectors.
f64 tolerance: 1e-12 absolute (observed ~5e-16; the kernel folds fixed joints and uses
c I + s K + (1-c) a a^T instead of I + s K + (1-c) K^2, so it is not bit-identical to the oracle)."""
import ctypes as C

import numpy as np
import pytest

import optas_amd
from conftest import KUKA_KIN, MED7_KIN, SEED, TESTER_KIN, oh_debug
from optas_amd import _lib
from optas_amd.models import KinematicsHandle, RobotModel
from oracle.robot import OracleRobot

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.mark.parametrize("tag,kin", [("kuka_lwr", KUKA_KIN), ("kuka_lwr_mid", KUKA_KIN), ("med7", MED7_KIN), ("tester", TESTER_KIN)])
def test_golden_vectors(hip_lib, golden_fk, tag, kin):
    robot = RobotModel(urdf_filename=kin)
    link = str(golden_fk[f"{tag}_link"])
    Q = golden_fk[f"{tag}_q"]
    pose, J = robot._kin(link).fk_jac(Q)
    assert np.abs(pose - golden_fk[f"{tag}_pose"]).max() < TOL  # position and reference-signed quaternion
    assert np.abs(J - golden_fk[f"{tag}_J"]).max() < TOL
    if tag == "kuka_lwr_mid":
        assert not J[:, :, 4:].any()  # joints past the link: zero columns (models.py:1251-1254)


def test_reference_shaped_api(hip_lib, golden_fk):
    robot = RobotModel(urdf_filename=KUKA_KIN)
    link = "end_effector_ball"
    Q = golden_fk["kuka_lwr_q"]
    # single configuration and ndof-by-n trajectory, like robot.get_global_link_position(link, q)
    p1 = robot.get_global_link_position(link, Q[0])
    assert p1.shape == (3,) and np.abs(p1 - golden_fk["kuka_lwr_pose"][0, :3]).max() < TOL
    P = robot.get_global_link_position_function(link, n=len(Q))(Q.T)
    assert P.shape == (3, len(Q)) and np.abs(P.T - golden_fk["kuka_lwr_pose"][:, :3]).max() < TOL
    quat = robot.get_global_link_quaternion(link, Q.T)
    assert np.abs(quat.T - golden_fk["kuka_lwr_pose"][:, 3:]).max() < TOL
    Jl = robot.get_global_link_geometric_jacobian(link, Q.T[:, :3])
    assert isinstance(Jl, list) and len(Jl) == 3 and np.abs(Jl[2] - golden_fk["kuka_lwr_J"][2]).max() < TOL
    R = robot.get_global_link_rotation(link, Q[5])
    assert np.abs(R - OracleRobot(KUKA_KIN).get_global_link_rotation(link, Q[5])).max() < 1e-12
    qN = optas_amd.deg2rad([0, 45, 0, -90, 0, -45, 0])
    assert np.allclose(robot.get_global_link_position(link, qN), [-0.868914357137, 0.0, 0.317071067812], atol=1e-11)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 5000])
def test_ragged_sizes_and_layouts(hip_lib, n):
    robot = RobotModel(urdf_filename=KUKA_KIN)
    orc = OracleRobot(KUKA_KIN)
    link = "end_effector_ball"
    rng = np.random.default_rng(SEED + n)
    Q = rng.uniform(-2.9, 2.9, (n, 7))
    kin = robot._kin(link)
    pose, J = kin.fk_jac(Q)
    idx = rng.choice(n, min(n, 16), replace=False)
    for i in idx:
        assert np.abs(pose[i, :3] - orc.get_global_link_position(link, Q[i])).max() < TOL
        assert np.abs(pose[i, 3:] - orc.get_global_link_quaternion(link, Q[i])).max() < TOL
        assert np.abs(J[i] - orc.get_global_link_geometric_jacobian(link, Q[i])).max() < TOL
    # structure-of-arrays device variant gives the same numbers
    lib = _lib.load()
    dq = _lib.DeviceBuffer(Q.nbytes).upload(np.ascontiguousarray(Q.T))
    dp, dJ = _lib.DeviceBuffer(n * 7 * 8), _lib.DeviceBuffer(n * 42 * 8)
    _lib.check(lib.oh_fk_jac_soa_device(kin._h, n, dq.ptr, dp.ptr, dJ.ptr), "soa")
    assert np.array_equal(dp.download(np.float64, (7, n)).T, pose)
    assert np.array_equal(dJ.download(np.float64, (42, n)).T.reshape(n, 6, 7), J)
    # pose-only / J-only calls
    pose2, none = kin.fk_jac(Q, want_jac=False)
    assert none is None and np.array_equal(pose2, pose)


def test_size_independent_properties(hip_lib):
    """At BASELINE scale (10^6 units): unit quaternions, J angular columns unit, J vs finite differences."""
    robot = RobotModel(urdf_filename=KUKA_KIN)
    kin = robot._kin("end_effector_ball")
    rng = np.random.default_rng(SEED)
    n = 1_000_000
    Q = rng.uniform(-2.9, 2.9, (n, 7))
    pose, J = kin.fk_jac(Q)
    assert np.abs(np.linalg.norm(pose[:, 3:], axis=1) - 1.0).max() < 1e-14
    assert np.abs(np.linalg.norm(J[:, 3:, :], axis=1) - 1.0).max() < 1e-14
    h = 1e-6
    for j in (0, 3, 6):
        d = np.zeros(7)
        d[j] = h
        pp, _ = kin.fk_jac(Q[:4096] + d, want_jac=False)
        pm, _ = kin.fk_jac(Q[:4096] - d, want_jac=False)
        assert np.abs((pp[:, :3] - pm[:, :3]) / (2 * h) - J[:4096, :3, j]).max() < 1e-8
    assert np.linalg.norm(pose[:, :3], axis=1).max() < 1.4105  # 0.11+4*0.2+0.19+0.078+0.2323: LWR + tool fully stretched


def test_errors(hip_lib):
    lib = _lib.load()
    robot = RobotModel(urdf_filename=KUKA_KIN)
    d = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_KINEMATICS, ndof=7)
    h = C.c_void_p()
    _lib.check(lib.oh_create(C.byref(d), C.byref(h)), "create")
    q = np.zeros((1, 7))
    assert lib.oh_fk_jac(h, 1, _lib._ptr(q), None, None) == 3  # OH_ERR_STATE: constants not set
    bad = robot.kinematic_chain("end_effector_ball")
    bad.jtype[0] = 5
    assert lib.oh_set_constants(h, C.byref(bad)) == 1 and b"joint type" in lib.oh_last_error()
    assert lib.oh_fk_jac(h, 0, _lib._ptr(q), None, None) == 1
    assert lib.oh_solve(h, 1, _lib._ptr(q), _lib._ptr(q), None, None, None, None, None) == 3  # kinematics-only handle
    lib.oh_destroy(h)


def test_transforms_and_base_frame_variants_match_oracle(hip_lib):
    """models.py:826-898, 949-1023, 1108-1122, 1320-1344 through the product's RobotModel (forward kinematics on the GPU) against
    the oracle's literal restatement; T_L invt(T_B) is the convention the reference's tests pin (tests/test_models.py:505-511)."""
    import numpy as np

    from conftest import KUKA_KIN, SEED
    from optas_amd.models import RobotModel
    from oracle.robot import OracleRobot

    rm, ro = RobotModel(urdf_filename=KUKA_KIN), OracleRobot(KUKA_KIN)
    rng = np.random.default_rng(SEED)
    for _ in range(4):
        q = rng.uniform(-1.5, 1.5, 7)
        for link, base in (("end_effector_ball", "lwr_arm_3_link"), ("lwr_arm_5_link", "lwr_arm_7_link"), ("lwr_arm_6_link", ro.get_root())):
            assert np.abs(rm.get_global_link_transform(link, q) - ro.get_global_link_transform(link, q)).max() < 1e-13
            assert np.abs(rm.get_link_transform(link, q, base) - ro.get_link_transform(link, q, base)).max() < 1e-13
            assert np.abs(rm.get_link_position(link, q, base) - ro.get_link_position(link, q, base)).max() < 1e-13
            assert np.abs(rm.get_link_rotation(link, q, base) - ro.get_link_rotation(link, q, base)).max() < 1e-13
            qa, qb = rm.get_link_quaternion(link, q, base), ro.get_link_quaternion(link, q, base)
            assert np.abs(qa - qb).max() < 1e-13  # same sign as the reference's chain
            assert np.abs(rm.get_link_geometric_jacobian(link, q, base) - ro.get_link_geometric_jacobian(link, q, base)).max() < 1e-13
            assert np.abs(rm.get_link_linear_jacobian(link, q, base) - ro.get_link_geometric_jacobian(link, q, base)[:3]).max() < 1e-13
        assert np.abs(rm.get_link_position_function("lwr_arm_5_link", ro.get_root())(q) - ro.get_global_link_position("lwr_arm_5_link", q)).max() < 1e-13


@pytest.mark.parametrize("specialize", ["0", "1"])
@pytest.mark.parametrize("ndof,n_chain", [(12, 12), (10, 7), (16, 16), (3, 2)])
def test_reference_layout_staging_for_long_and_partial_chains(hip_lib, monkeypatch, ndof, n_chain, specialize):
    """The reference-layout kernel stages q, pose and J through LDS (a tile of 128 units x 6 ndof doubles: beyond 8 joints it needs more than
    the default 48 KB of dynamic LDS): a synthetic chain of up to 16 joints, with and without joints off the chain, must give what the SoA
    kernel -- no LDS, pinned against the oracle above -- gives, for block-ragged sizes."""
    oh_debug(monkeypatch, specialize=specialize)  # the generic kernels and the ones compiled for this very chain
    base = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain("end_effector_ball")
    ch = _lib.oh_chain()
    C.memmove(C.byref(ch), C.byref(base), C.sizeof(ch))
    ch.ndof, ch.n_chain = ndof, n_chain
    for k in range(n_chain):  # repeat the LWR's joints; the chain uses every other model joint when there are spare ones
        s = k % 7
        ch.jtype[k], ch.axcode[k], ch.r0ident[k] = base.jtype[s], base.axcode[s], base.r0ident[s]
        ch.qidx[k] = k if ndof == n_chain else min(ndof - 1, k + (k >= 3) * (ndof - n_chain))
        for i in range(9):
            ch.R0[k][i] = base.R0[s][i]
        for i in range(3):
            ch.p0[k][i], ch.axis[k][i] = base.p0[s][i], base.axis[s][i]
        for i in range(4):
            ch.quat0[k][i] = base.quat0[s][i]
    kin = KinematicsHandle(ch)
    lib = _lib.load()
    rng = np.random.default_rng(SEED + ndof)
    for n in (1, 127, 129, 300, 4097):
        Q = rng.uniform(-2.0, 2.0, (n, ndof))
        pose, J = kin.fk_jac(Q)
        dq = _lib.DeviceBuffer(Q.nbytes).upload(np.ascontiguousarray(Q.T))
        dp, dJ = _lib.DeviceBuffer(n * 7 * 8), _lib.DeviceBuffer(n * 6 * ndof * 8)
        _lib.check(lib.oh_fk_jac_soa_device(kin._h, n, dq.ptr, dp.ptr, dJ.ptr), "soa")
        # (the two layouts are two instantiations of one function body: the compiler may contract their products differently)
        assert np.abs(dp.download(np.float64, (7, n)).T - pose).max() <= 4e-15
        assert np.abs(dJ.download(np.float64, (6 * ndof, n)).T.reshape(n, 6, ndof) - J).max() <= 4e-15
        off = [c for c in range(ndof) if c not in [ch.qidx[k] for k in range(n_chain)]]
        assert np.abs(J[:, :, off]).max() == 0.0 if off else True
        for b in (dq, dp, dJ):
            b.free()
