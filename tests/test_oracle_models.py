"""Pins the oracle's RobotModel restatement with the known answers the reference's own tests hold
(tests/test_models.py:206-250 names/indexes/limits, :399-428 origins/axes, :461 reach, :505-511
base-frame convention) and with closed forms / finite differences."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from conftest import KUKA_KIN, MED7_KIN, TESTER_KIN
from oracle.robot import OracleRobot
from oracle.structured import FoldedChain


@pytest.fixture(scope="module")
def tester():
    return OracleRobot(TESTER_KIN)


def test_tester_robot_bookkeeping(tester):
    assert tester.joint_names == ["joint0", "joint1", "joint2", "eff_joint"]
    assert tester.link_names == ["world", "link1", "link2", "link3", "eff"]
    assert tester.actuated_joint_names == ["joint0", "joint1", "joint2"]
    assert tester.ndof == 3
    assert [tester.get_actuated_joint_index(n) for n in tester.actuated_joint_names] == [0, 1, 2]
    assert np.allclose(tester.lower_actuated_joint_limits, [-1e9, -1, 0])
    assert np.allclose(tester.upper_actuated_joint_limits, [1e9, 1, 1])
    assert np.allclose(tester.velocity_actuated_joint_limits, [1e9, 1, 1])
    origins = [([0, 0, 0], [0, 0, 0]), ([2, 0, 0], [0, 0, 0]), ([1, 0, 0], [0, 0, 0]), ([0, 0, 0.5], [0, 0, 0])]
    for j, (xyz, rpy) in zip(tester.joints, origins):
        a, b = tester.get_joint_origin(j)
        assert np.allclose(a, xyz) and np.allclose(b, rpy)
    for j in tester.joints[:-1]:
        assert np.allclose(tester.get_joint_axis(j), [0, 0, 1])
    assert np.allclose(tester.get_joint_axis(tester.joints[-1]), [1, 0, 0])  # default axis, models.py:658
    assert tester.get_root() == "world"
    assert tester.get_chain("world", "eff") == ["joint0", "joint1", "joint2", "eff_joint"]


def test_tester_robot_fk_closed_form(tester):
    # Rz(q0) Tx(2) Rz(q1) Tx(1) Tz(q2) Tz(.5)
    rng = np.random.default_rng(1)
    for _ in range(100):
        q = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-1, 1), rng.uniform(0, 1)])
        p = tester.get_global_link_position("eff", q)
        expect = [2 * np.cos(q[0]) + np.cos(q[0] + q[1]), 2 * np.sin(q[0]) + np.sin(q[0] + q[1]), q[2] + 0.5]
        assert np.allclose(p, expect, atol=1e-14)
        assert np.linalg.norm(p) <= np.linalg.norm([3, 0, 1.5]) + 1e-12  # tests/test_models.py:461
        R = tester.get_global_link_rotation("eff", q)
        assert np.allclose(R, Rot.from_euler("z", q[0] + q[1]).as_matrix(), atol=1e-14)
    assert np.allclose(tester.get_global_link_transform("world", [0.1, 0.2, 0.3]), np.eye(4))
    # SURVEY App. D value
    assert np.allclose(tester.get_global_link_position("eff", [0.3, -0.4, 0.7]), [2.905677143529, 0.491206996676, 1.2], atol=1e-11)


def test_base_frame_convention(tester):
    # T_L . invt(T_B), not invt(T_B) . T_L  (models.py:896-898; tests/test_models.py:505-511)
    q = np.array([0.4, -0.3, 0.6])
    TL = tester.get_global_link_transform("eff", q)
    TB = tester.get_global_link_transform("link2", q)
    assert np.allclose(tester.get_link_transform("eff", q, "link2"), TL @ np.linalg.inv(TB), atol=1e-14)
    assert np.allclose(tester.get_link_transform("eff", q, "world"), TL)


@pytest.mark.parametrize("kin,link", [(KUKA_KIN, "end_effector_ball"), (MED7_KIN, "lbr_link_ee"), (TESTER_KIN, "eff")])
def test_quaternion_and_jacobian(kin, link):
    r = OracleRobot(kin)
    rng = np.random.default_rng(2)
    lo, up = np.maximum(r.lower_actuated_joint_limits, -3), np.minimum(r.upper_actuated_joint_limits, 3)
    for _ in range(10):
        q = rng.uniform(lo, up)
        quat = r.get_global_link_quaternion(link, q)
        ref = Rot.from_matrix(r.get_global_link_rotation(link, q)).as_quat()
        assert min(np.abs(quat - ref).max(), np.abs(quat + ref).max()) < 1e-13
        J = r.get_global_link_geometric_jacobian(link, q)
        h = 1e-6
        for j in range(r.ndof):
            d = np.zeros(r.ndof)
            d[j] = h
            Jn = (r.get_global_link_position(link, q + d) - r.get_global_link_position(link, q - d)) / (2 * h)
            assert np.allclose(J[:3, j], Jn, atol=1e-8)
            Rp, Rm = r.get_global_link_rotation(link, q + d), r.get_global_link_rotation(link, q - d)
            W = (Rp - Rm) / (2 * h) @ r.get_global_link_rotation(link, q).T
            assert np.allclose(J[3:, j], [W[2, 1], W[0, 2], W[1, 0]], atol=1e-8)
            qn = (r.get_global_link_quaternion(link, q + d) - r.get_global_link_quaternion(link, q - d)) / (2 * h)
            assert np.allclose(r.quaternion_jacobian(link, q)[:, j], qn, atol=1e-8)


def test_known_poses():
    k = OracleRobot(KUKA_KIN)
    qN = np.deg2rad([0, 45, 0, -90, 0, -45, 0])
    assert np.allclose(k.get_global_link_position("end_effector_ball", qN), [-0.868914357137, 0.0, 0.317071067812], atol=1e-11)
    m = OracleRobot(MED7_KIN)
    assert np.allclose(m.get_global_link_position("lbr_link_ee", np.deg2rad([0, 30, 0, -90, 0, -30, 0])), [0.672410161514, 0, 0.486410161514], atol=1e-11)
    assert k.ndof == 7 and m.ndof == 7
    assert k.get_root() == "lwr_arm_0_link" and len(k.get_chain(k.get_root(), "end_effector_ball")) == 12


def test_golden_fixture_matches_oracle(golden_fk):
    for tag, kin in (("kuka_lwr", KUKA_KIN), ("kuka_lwr_mid", KUKA_KIN), ("med7", MED7_KIN), ("tester", TESTER_KIN)):
        r = OracleRobot(kin)
        link = str(golden_fk[f"{tag}_link"])
        Q = golden_fk[f"{tag}_q"]
        for i in range(0, len(Q), 7):
            assert np.allclose(r.get_global_link_position(link, Q[i]), golden_fk[f"{tag}_pose"][i, :3], atol=1e-15)
            assert np.allclose(r.get_global_link_geometric_jacobian(link, Q[i]), golden_fk[f"{tag}_J"][i], atol=1e-15)


def test_folded_chain_equals_literal_chain():
    k = OracleRobot(KUKA_KIN)
    fc = FoldedChain(k, "end_effector_ball")
    rng = np.random.default_rng(3)
    Q = rng.uniform(-2.9, 2.9, (20, 7))
    e, Re, Jp, Jw = fc.jac(Q)
    for i in range(20):
        J = k.get_global_link_geometric_jacobian("end_effector_ball", Q[i])
        assert np.allclose(e[i], k.get_global_link_position("end_effector_ball", Q[i]), atol=1e-14)
        assert np.allclose(Re[i], k.get_global_link_rotation("end_effector_ball", Q[i]), atol=1e-14)
        assert np.allclose(Jp[i], J[:3], atol=1e-14) and np.allclose(Jw[i], J[3:], atol=1e-14)


def test_add_base_frame():
    k = OracleRobot(KUKA_KIN)
    q = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    p0 = k.get_global_link_position("end_effector_ball", q)
    k.add_base_frame("global_world", xyz=[0, -0.25, 0])
    assert k.get_root() == "global_world"
    assert np.allclose(k.get_global_link_position("end_effector_ball", q), p0 + [0, -0.25, 0], atol=1e-15)


def test_rnea_oracle_physics():
    """The literal RNEA restatement (models.py:1731-1884) is physically consistent: symmetric PD mass matrix,
    affine in qdd, gravity torque = dV/dq (V from the FK oracle).  The reference's own check is against pybullet
    at atol 8e-2 (tests/test_models.py:1040-1052); pybullet is absent here."""
    import os

    from conftest import GOLDEN
    from oracle.robot import rnea, rnea_tables

    for kin in (MED7_KIN, os.path.join(GOLDEN, "tester_robot_revolute.kin.json")):
        r = OracleRobot(kin)
        n = r.ndof
        rng = np.random.default_rng(4)
        q, qd, qdd = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        tau = rnea(r, q, qd, qdd)
        t0 = rnea(r, q, qd, np.zeros(n))
        M = np.stack([rnea(r, q, qd, np.eye(n)[i]) - t0 for i in range(n)], 1)
        assert np.abs(M - M.T).max() < 1e-13 and np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0
        assert np.abs(t0 + M @ qdd - tau).max() < 1e-12
        m, cm, _, _, _, _ = rnea_tables(r)
        links = [l for l in r.links if r.link_inertials[l] is not None][1:]
        V = lambda qq: sum(m[i] * 9.81 * (r.get_global_link_transform(l, qq)[:3, :3] @ cm[:, i] + r.get_global_link_transform(l, qq)[:3, 3])[2] for i, l in enumerate(links))
        g = rnea(r, q, np.zeros(n), np.zeros(n))
        h = 1e-6
        gn = np.array([(V(q + h * np.eye(n)[i]) - V(q - h * np.eye(n)[i])) / (2 * h) for i in range(n)])
        assert np.abs(g - gn).max() < 1e-6
    with pytest.raises(NotImplementedError):
        rnea(OracleRobot(KUKA_KIN), np.zeros(7), np.zeros(7), np.zeros(7))  # first joint not fixed (models.py:1748)


def test_quaternion_batch_equals_the_scalar_chain_walk():
    """oracle/robot.py:quaternion_batch (the vectorised form the bench-scale GPU tests check every knot with) against the scalar restatement of
    models.py:1049-1088, on robots with fixed, revolute and prismatic joints."""
    for kin, link in ((KUKA_KIN, "end_effector_ball"), (MED7_KIN, None), (TESTER_KIN, "eff")):
        r = OracleRobot(kin)
        link = link or r.link_names[-1]
        Q = np.random.default_rng(7).uniform(-2.9, 2.9, (40, r.ndof))
        qb = r.quaternion_batch(link, Q)
        assert qb.shape == (40, 4)
        for i in range(40):
            assert np.abs(qb[i] - r.get_global_link_quaternion(link, Q[i])).max() <= 1e-15
    assert np.array_equal(OracleRobot(TESTER_KIN).quaternion_batch("world", np.zeros((2, 3))), np.tile([0.0, 0, 0, 1], (2, 1)))
