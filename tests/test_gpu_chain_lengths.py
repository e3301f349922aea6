"""Chain lengths other than the KUKA's (round-4 verdict, Missing 4): the reference's RobotModel takes any URDF (models.py:233-321, 332-550) and ships
planar_3dof.urdf and the tester robots; until round 4 the structured kernel families were instantiated for 6 and 7 joints only.  Round 5: the IK and
the position-tracking family for 2 ... 8 actuated joints, the orientation-locked family for 4 ... 8.  Robots: tests/golden/planar_3dof.kin.json (3
revolute), tests/golden/tester_robot.kin.json (continuous, revolute, prismatic), and the KUKA LWR cut after its 4th / 5th joint or extended by an 8th.
Every answer is compared with the numpy port of the same state machine and graded on the literal NLP (oracle/problems.py) by kkt_reference_form."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, KUKA_KIN, SEED, TESTER_KIN
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend, IKBackend
from optas_amd.models import RobotModel
from oracle.ik_al import solve_ik_al
from oracle.problems import DualArmNLP, FigureEightNLP, IKExampleNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain, StructuredFigureEight, solve_free_lm, solve_structured_lm

pytestmark = pytest.mark.gpu
PLANAR_KIN = os.path.join(GOLDEN, "planar_3dof.kin.json")


def _kuka_variant(tmp_path, n):
    """kuka_lwr.kin.json with n actuated joints: cut after joint n (a 10 cm tool on the last link), or -- n = 8 -- a wrist joint added behind the flange."""
    d = json.load(open(KUKA_KIN))
    joints = {j["name"]: j for j in d["joints"]}
    out = copy.deepcopy(d)
    out["name"] = f"kuka{n}"
    if n < 7:
        keep = [f"lwr_arm_{i}_joint" for i in range(n)]
        last = joints[keep[-1]]["child"]
        out["joints"] = [joints[k] for k in keep] + [{"name": "tool_joint", "type": "fixed", "parent": last, "child": "tool", "xyz": [0.0, 0.0, 0.1], "rpy": [0.0, 0.0, 0.0]}]
        names = {"lwr_arm_0_link", "tool"} | {joints[k]["child"] for k in keep}
        out["links"] = [l for l in d["links"] if l["name"] in names] + [{"name": "tool"}]
    else:
        js = []
        for j in d["joints"]:
            if j["name"] == "lwr_arm_7_joint":  # the fixed flange joint: a wrist roll about y takes its place, the flange follows
                js.append({"name": "wrist_extra_joint", "type": "revolute", "parent": "lwr_arm_7_link", "child": "wrist_extra_link", "xyz": [0.0, 0.0, 0.05],
                           "rpy": [0.0, 0.0, 0.0], "axis": [0.0, 1.0, 0.0], "limit": {"lower": -2.0, "upper": 2.0, "velocity": 2.0, "effort": 50.0}})
                js.append({**j, "parent": "wrist_extra_link"})
            else:
                js.append(j)
        out["joints"] = js
        out["links"] = d["links"] + [{"name": "wrist_extra_link"}]
    path = os.path.join(str(tmp_path), f"kuka{n}.kin.json")
    json.dump(out, open(path, "w"))
    return path, ("tool" if n < 7 else "end_effector_ball")


def _robots(tmp_path):
    """(tag, kin file, link, nominal configuration)"""
    out = [("planar3", PLANAR_KIN, "end", np.array([0.3, -0.5, 0.4])), ("tester3", TESTER_KIN, "eff", np.array([0.4, -0.3, 0.5]))]
    for n in (4, 5, 8):
        path, link = _kuka_variant(tmp_path, n)
        out.append((f"kuka{n}", path, link, np.deg2rad([0, 30, 0, -90, 0, -30, 0, 20])[:n]))
    return out


def test_inverse_kinematics_for_three_to_eight_joints(hip_lib, tmp_path):
    rng = np.random.default_rng(SEED + 71)
    for tag, kin, link, qn in _robots(tmp_path):
        orc = OracleRobot(kin)
        n = orc.ndof
        assert n == len(qn)
        ik, ch = IKExampleNLP(orc, link), FoldedChain(orc, link)
        lo, up = np.maximum(ik.lo, -3.0), np.minimum(ik.up, 3.0)  # (the tester robot's continuous joint has the reference's +-1e9 default)
        ik.lo, ik.up = lo, up
        chain = RobotModel(urdf_filename=kin).kinematic_chain(link)
        assert chain.ndof == n
        be = IKBackend(chain, lo, up, max_iter=300)
        B = 64
        q0 = np.clip(qn + rng.uniform(-0.2, 0.2, (B, n)), lo, up)
        pg, _, _, _ = ch.fk(np.clip(q0 + rng.uniform(-0.3, 0.3, (B, n)), lo, up))  # reachable by construction
        p = np.concatenate([q0, pg], 1)
        res = be.solve(q0, p)
        mu, zlo, zup = be.multipliers(B)
        ok = res.status == 0
        assert ok.mean() >= 0.9, (tag, np.bincount(res.status))
        e2, _, _, _ = ch.fk(res.x[ok])
        assert np.abs(pg[ok] - e2).max() <= 1e-9 and (res.x >= lo).all() and (res.x <= up).all()
        for b in np.flatnonzero(ok)[:12]:
            r = solve_ik_al(ch, q0[b], q0[b], pg[b], lo, up, tol=1e-6, tol_feas=1e-9, max_iter=300)
            assert r["status"] == 0 and np.abs(res.x[b] - r["x"]).max() <= 1e-7 and abs(int(res.iters[b]) - r["iterations"]) <= 2, (tag, b)
            lam = np.concatenate([zlo[b], zup[b], np.maximum(mu[b], 0), np.maximum(-mu[b], 0)])
            assert np.abs(ik.df(res.x[b], p[b]) - ik.dv(res.x[b], p[b]).T @ lam).max() <= 1e-6 and abs(ik.f(res.x[b], p[b]) - res.f[b]) <= 1e-12
        print(f"IK {tag}: {n} joints, {int(ok.sum())}/{B} converged, steps p50 {int(np.median(res.iters))}")
        be.close()


def test_position_tracking_for_three_to_eight_joints(hip_lib, tmp_path):
    """dual_arm.py's problem per arm (position-only tracking, q_0 pinned, Euler rows eliminated) on the other robots: the GPU = the numpy port of the
    state machine, and a KKT point of the literal NLP (oracle/problems.py:DualArmNLP with the robot in both slots)."""
    rng = np.random.default_rng(SEED + 72)
    T = 24
    ts = np.linspace(0.0, 1.0, T)
    for tag, kin, link, qn in _robots(tmp_path):
        orc = OracleRobot(kin)
        n = orc.ndof
        ch = FoldedChain(orc, link)
        scale = 0.5 if tag.endswith("3") else 0.1
        offs = scale * np.stack([np.sin(np.pi * ts) * 0.8, ts, -0.5 * ts * (tag != "planar3")], 1)  # (T, 3); the planar arm stays in its plane
        dt = 10.0 / (T - 1)
        chain = RobotModel(urdf_filename=kin).kinematic_chain(link)
        be = FigureEightBackend(chain, T, dt, offs, w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False, path_in_frame=False)
        B = 8
        qc = qn + rng.uniform(-0.1, 0.1, (B, n))
        x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((B, n * (T - 1)))], 1)
        res = be.solve(x0, qc)
        assert (res.status == 0).all(), (tag, res.status)
        nlp = DualArmNLP(orc, OracleRobot(kin), link=link, T=T)
        nlp.offsets = {"l": offs.T, "r": offs.T}
        for b in range(0, B, 2):
            s = solve_free_lm(ch, T, dt, offs, qc[b], Q0=np.tile(qc[b], (T, 1)), max_iter=400)
            assert s["status"] == 0 and abs(s["f"] - res.f[b]) <= 1e-9 * max(1.0, s["f"]) and abs(s["iters"] - int(res.iters[b])) <= max(2, s["iters"] // 20), (tag, b, s["f"], res.f[b])
            x2, p2 = np.concatenate([res.x[b], res.x[b + 1]]), np.concatenate([qc[b], qc[b + 1]])
            assert abs(nlp.f(x2, p2) - (res.f[b] + res.f[b + 1])) <= 1e-10 and np.abs(nlp.a(x2, p2)).max() <= 1e-12
            k = kkt_reference_form(nlp, x2, p2)
            assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-12, (tag, k)
        print(f"tracking {tag}: {n} joints, steps {res.iters.tolist()}")
        be.close()


def test_orientation_locked_family_for_four_five_and_eight_joints(hip_lib, tmp_path):
    """figure_eight_plan.py's problem (orientation rows h = quat_c - quat(q_t), null space of n - 3 dimensions per knot) on chains of 4, 5 and 8
    joints: through the persistent kernel (small batch) and through the batched launches, against the numpy port and the literal NLP."""
    rng = np.random.default_rng(SEED + 73)
    T = 20
    for tag, kin, link, qn in _robots(tmp_path)[2:]:
        orc = OracleRobot(kin)
        n = orc.ndof
        prob = StructuredFigureEight(orc, link, T=T, Tmax=4.0)
        prob.local_path = prob.local_path * (0.25 if n < 6 else 1.0)  # four joints leave ONE degree of freedom per knot: a short path
        prob.local_path[:, 2] = 0.6 * prob.local_path[:, 0]  # (... and that one moves the tool of the cut KUKA along the local z axis only)
        nlp = FigureEightNLP(orc, link, T=T, Tmax=4.0)
        nlp.local_path = prob.local_path.T.copy()
        chain = RobotModel(urdf_filename=kin).kinematic_chain(link)
        B = 8
        qc = qn + rng.uniform(-0.05, 0.05, (B, n))
        x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((B, n * (T - 1)))], 1)
        out = {}
        for mode, opts in (("tail", {}), ("batched", {"tail_threshold": 0, "compaction": 0})):
            be = FigureEightBackend(chain, T, prob.dt, prob.local_path, max_iter=400, tol=1e-6).set_options(opts)
            out[mode] = be.solve(x0, qc)
            be.close()
            assert (out[mode].status == 0).all(), (tag, mode, out[mode].status)
        assert np.abs(out["tail"].f - out["batched"].f).max() <= 1e-8 * max(1.0, np.abs(out["tail"].f).max())
        res = out["batched"]
        for b in range(0, B, 3):
            s = solve_structured_lm(prob, qc[b], max_iter=400, tol=1e-6)
            assert s["status"] == 0 and abs(s["f"] - res.f[b]) <= 1e-8 * max(1.0, s["f"]), (tag, b, s["f"], res.f[b])
            x = res.x[b]
            assert abs(nlp.f(x, qc[b]) - res.f[b]) <= 1e-10 * max(1.0, res.f[b]) and np.abs(nlp.a(x, qc[b])).max() <= 1e-12 and np.abs(nlp.h(x, qc[b])).max() <= 1e-9
            k = kkt_reference_form(nlp, x, qc[b])
            assert k["stationarity"] <= 1e-4 and k["feasibility"] <= 1e-9, (tag, k)
        print(f"figure-eight {tag}: {n} joints (null space {n - 3}), steps tail {out['tail'].iters.tolist()} batched {res.iters.tolist()}")


def test_velocity_limited_figure_eight_on_five_and_eight_joints(hip_lib, tmp_path):
    """Inequality rows on the orientation-locked family for the other chain lengths (k_tail_vel / k_eval_lg instantiated for 4 ... 8 joints): joint-velocity
    limits that bind, against the numpy port of the state machine, rows checked on the returned trajectory."""
    rng = np.random.default_rng(SEED + 74)
    T = 20
    for tag, kin, link, qn in [r for r in _robots(tmp_path) if r[0] in ("kuka5", "kuka8")]:
        orc = OracleRobot(kin)
        n = orc.ndof
        prob = StructuredFigureEight(orc, link, T=T, Tmax=4.0)
        prob.local_path = prob.local_path * (0.25 if n < 6 else 1.0)
        prob.local_path[:, 2] = 0.6 * prob.local_path[:, 0]
        chain = RobotModel(urdf_filename=kin).kinematic_chain(link)
        qc = qn + rng.uniform(-0.05, 0.05, (4, n))
        x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((4, n * (T - 1)))], 1)
        free = FigureEightBackend(chain, T, prob.dt, prob.local_path, max_iter=400, tol=1e-6)
        rf = free.solve(x0, qc)
        free.close()
        vmax = 0.6 * np.abs(rf.x[:, n * T :]).max()  # below what the unconstrained plans use: the rows bind
        g = _lib.oh_guards()
        g.vel_limits = 1
        for j in range(n):
            g.dq_lo[j], g.dq_up[j] = -vmax, vmax
        be = FigureEightBackend(chain, T, prob.dt, prob.local_path, max_iter=600, tol=1e-6, guards=g)
        r = be.solve(x0, qc)
        be.close()
        assert (r.status == 0).all(), (tag, r.status)
        dQ = r.x[:, n * T :]
        assert np.abs(dQ).max() <= vmax + 1e-8 and (np.abs(dQ).max(1) >= vmax - 1e-6).all() and (r.f > rf.f).all()
        for b in (0, 3):
            s = solve_structured_lm(prob, qc[b], max_iter=600, tol=1e-6, vlimits=(np.full(n, -vmax), np.full(n, vmax)))
            assert s["status"] == 0 and abs(s["f"] - r.f[b]) <= 1e-7 * max(1.0, s["f"]), (tag, b, s["f"], r.f[b])
        print(f"velocity-limited figure-eight {tag}: f free {rf.f.round(4).tolist()} limited {r.f.round(4).tolist()}, steps {r.iters.tolist()}")


def test_chain_lengths_outside_the_instantiated_range_are_refused(hip_lib):
    import ctypes as C

    lib = _lib.load()
    h = C.c_void_p()
    lp = (C.c_double * 30)()
    for ndof, lock, ok in ((1, 0, False), (9, 0, False), (3, 1, False), (3, 0, True), (4, 1, True), (8, 1, True)):
        d = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_FIGURE_EIGHT, T=10, ndof=ndof, dt=0.1, w_path=1.0, w_vel=0.01, local_path=lp, lock_orientation=lock, hessian=0)
        rc = lib.oh_create(C.byref(d), C.byref(h))
        assert (rc == 0) == ok, (ndof, lock, rc, lib.oh_last_error())
        if rc == 0:
            lib.oh_destroy(h)
