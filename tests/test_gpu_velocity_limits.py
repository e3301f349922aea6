"""Joint-velocity limits on config 2 -- ``builder.enforce_model_limits(name, time_deriv=1)`` (builder.py:471-509), the first constraint a user of
figure_eight_plan.py adds: the script's optimum runs joint 0 at 2.01 rad/s against the KUKA LWR's 1.92.  Rows dq_t - vlo >= 0, vup - dq_t >= 0
on dq_t = (q_{t+1} - q_t)/dt couple neighbouring knots; the kernels treat them through the augmented Lagrangian inside k_couple_vel.
Checked against the numpy port (same state machine), against the reference-form KKT conditions on the literal 693-variable layout with its
686 extra k rows (oracle.problems.LimitedFigureEightNLP), and for the limits themselves."""
import os
import sys

import numpy as np
import pytest

from conftest import KUKA_KIN, SEED, oh_debug
from oracle.problems import LimitedFigureEightNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import StructuredFigureEight, solve_structured_lm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.figure_eight_plan import setup_solver  # noqa: E402

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"
QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])


@pytest.mark.parametrize("vmax", [None, 1.0])
def test_velocity_limited_figure_eight(hip_lib, vmax):
    orc = OracleRobot(KUKA_KIN)
    vl = np.asarray(orc.velocity_actuated_joint_limits) if vmax is None else np.full(7, vmax)
    kuka, solver = setup_solver(velocity_limits=True if vmax is None else (-vl, vl), solver_options={"max_iter": 600, "tol": 1e-7})
    assert solver.opt.nk == 2 * 7 * 49 and solver.opt.nv == 1114 + 686
    rng = np.random.default_rng(SEED + 3)
    qcs = QC0[None] + np.concatenate([np.zeros((1, 7)), rng.uniform(-0.05, 0.05, (3, 7))])
    B = len(qcs)
    solver.reset_parameters_batch({"qc": qcs})
    solver.reset_initial_seed_batch({"kuka/q/x": np.stack([np.tile(q.reshape(-1, 1), (1, 50)) for q in qcs])})
    sols = solver.solve_batch()
    st = solver.stats()
    assert st["success"]
    prob = StructuredFigureEight(orc, LINK, T=50)
    nlp = LimitedFigureEightNLP(orc, LINK, vlo=-vl, vup=vl, T=50)
    lam = solver.backend.multipliers(B)
    assert lam.shape == (B, 50, 14) and lam.min() >= 0.0
    for b in range(B):
        x = solver.opt.decision_variables.dict2vec(sols[b])
        dQ = np.asarray(sols[b]["kuka/dq"])
        assert np.abs(dQ).max(1).max() <= vl.max() + 1e-8 and np.all(np.abs(dQ).max(1) <= vl + 1e-8)
        assert np.abs(nlp.a(x, qcs[b])).max() <= 1e-12 and np.abs(nlp.h(x, qcs[b])).max() <= 1e-9 and nlp.k(x, qcs[b]).min() >= -1e-8
        assert abs(nlp.f(x, qcs[b]) - st["f"][b]) <= 1e-9 * st["f"][b]
        k = kkt_reference_form(nlp, x, qcs[b], active_tol=1e-7)
        assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-8 and k["complementarity"] <= 1e-6, (b, k)
        s = solve_structured_lm(prob, qcs[b], max_iter=600, tol=1e-7, vlimits=(-vl, vl))
        assert s["status"] == 0 and abs(s["f"] - st["f"][b]) <= 1e-8 * s["f"], (b, s["f"], st["f"][b])
        # same state machine (ratio test, line search on a rejected step, damping rule); a run of 100+ steps with 38 rows active parts from
        # the port's at one borderline ratio test and may end dozens of steps apart -- the optimum and the multipliers (below) do not
        assert abs(int(st["iterations"][b]) - s["iters"]) <= (max(3, s["iters"] // 4) if s["iters"] <= 100 else s["iters"] // 2), (b, st["iterations"][b], s["iters"])
        # multipliers in the reference's row order: [dq_t - vlo; vup - dq_t] at knot t
        lv = np.zeros((50, 14))
        lv[:49] = s["lam_v"]
        assert np.abs(lam[b] - lv).max() <= 1e-4 * max(1.0, lv.max())
    # the rows bind: with the robot's own limits in the nominal instance (its unconstrained optimum runs joint 0 at 2.01 rad/s), at 1 rad/s everywhere
    assert lam[0].max() > 0 and (vmax is None or (lam.max((1, 2)) > 0).all())


def test_velocity_limited_batch_is_compacted_invisibly(hip_lib, monkeypatch):
    """Orientation-locked family with velocity rows, a batch large enough to be compacted while it drains (round 2): the multipliers of the
    velocity rows move with the instance like those of the other rows; compaction on and off must end in the same points and multipliers."""
    oh_debug(monkeypatch, tail_vel="0")  # (the batched launches to the end: with the persistent kernel of round 3 a batch of 640 never sees a compaction)
    B = 640
    rng = np.random.default_rng(SEED + 43)
    qcs = QC0[None] + rng.uniform(-0.08, 0.08, (B, 7))
    seeds = np.stack([np.tile(q.reshape(-1, 1), (1, 50)) for q in qcs])
    out = {}
    for mode in ("0", "1"):
        oh_debug(monkeypatch, compaction=mode)
        kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-7})
        solver.reset_parameters_batch({"qc": qcs})
        solver.reset_initial_seed_batch({"kuka/q/x": seeds})
        solver.solve_batch(stacked=True)
        st = solver.stats()
        be = solver.backend
        out[mode] = (st["f"].copy(), st["iterations"].copy(), st["status"].copy(), st["solution"].x.copy(), be.multipliers(B), be.timing()["compactions"])
        be.close()
    (f0, i0, s0, x0, l0, c0), (f1, i1, s1, x1, l1, c1) = out["0"], out["1"]
    assert c0 == 0 and c1 >= 1
    assert (s0 == 0).all() and (s1 == 0).all()
    # a survivor restarts from its accepted point, which the restart re-retracts to the floor tolerance (far from the solution accepted points
    # keep up to 1e-5 of orientation violation): the paths part by rounding-size amounts and a long run may end some steps apart -- the optima
    # and their multipliers must not
    both = (s0 == 0) & (s1 == 0)
    # (which instances a restart catches mid-run depends on when the host looks at the running count: 0.97 with a look every iteration, 0.94 with
    #  the sparse looks small launches get since round 3 -- the bound is a statement about "most", the optima below are the invariant)
    assert (np.abs(i0.astype(int) - i1)[both] <= np.maximum(3, i0[both] // 4)).mean() >= 0.9
    assert np.abs(f0 - f1)[both].max() <= 1e-7 * np.abs(f0).max()
    same = both & (i0 == i1)
    assert same.mean() > 0.5 and np.abs(x0[same] - x1[same]).max() <= 1e-4  # ~1e-5 rad of play along the weakly curved elbow-swivel direction
    assert l0.shape == l1.shape and np.abs(l0[same] - l1[same]).max() <= 1e-3 * max(1.0, np.abs(l0).max())
    assert np.abs(l0[..., -14:]).max() > 0  # velocity rows are active somewhere (the LWR's limit of joint 0 binds on the nominal path)


def test_batch_of_16384_has_no_stalled_instance_and_matches_its_instances_solved_alone(hip_lib, monkeypatch):
    """Round-2 verdict, Weak 4 / Next 5: in a 16 384-instance velocity-limited batch (seed 5, the reproducer of DESIGN 7.4) one instance sat at a
    reduced gradient of 1e-5 until the cap of 600 while the same input alone converged in 64 steps; with compaction off three others did.
    Cause (reproduced in the numpy port, oracle/structured.py:retract_tol): in the end game of the outer loop the steps predict decreases of
    1e-12 while 1e-10 of orientation violation -- the retraction tolerance -- is worth 1.5e-10 of objective: steps were accepted or refused by
    the rounding of the retraction.  With the tolerance tied to the prediction no instance stalls, with compaction on or off, and every instance
    reaches the optimum it reaches alone."""
    B = 16384
    rng = np.random.default_rng(5)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    res = {}
    # "tail": the default -- such a batch drains in the persistent kernel k_tail_vel (one wavefront per instance, the whole outer loop on chip);
    # "batched": OH_TAIL_VEL=0, batched launches to the end with restart compactions; "plain": ... and without compaction
    for mode, env in (("tail", {}), ("batched", {"OH_TAIL_VEL": "0"}), ("plain", {"OH_TAIL_VEL": "0", "OH_COMPACTION": "0"})):
        for k in ("OH_TAIL_VEL", "OH_COMPACTION"):
            oh_debug(monkeypatch, **{k: None})
        for k, v in env.items():
            oh_debug(monkeypatch, **{k: v})
        kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
        x0 = np.zeros((B, solver.opt.nx))
        x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
        r = solver.solve_batch_arrays(x0, qcs)
        assert (r.status == 0).all(), (mode, np.flatnonzero(r.status != 0)[:8], r.kkt[r.status != 0][:8])
        assert r.iters.max() <= 300, (mode, r.iters.max())  # the slowest instance: 141 - 158 steps
        assert (r.kkt[:, 0] <= 1e-6).all() and (r.kkt[:, 1] <= 1e-9).all()
        tm = solver.backend.timing()
        lam_b = solver.backend.multipliers(B)
        assert (tm["tail_iterations"] > 0) == (mode == "tail") and (tm["compactions"] >= 5) == (mode == "batched"), (mode, tm)
        # an instance's iterates do not depend on the batch in the persistent kernel and in the plain batched path: bit-identical to the instance
        # solved alone (under the same settings); the four former cap hitters among the sample
        idx = np.concatenate([[14952, 1725, 2953, 3292], np.random.default_rng(6).choice(B, 28, replace=False)])
        if mode != "batched":
            for b in idx:
                a = solver.solve_batch_arrays(x0[b : b + 1], qcs[b : b + 1])
                assert a.status[0] == 0 and a.iters[0] == r.iters[b] and a.f[0] == r.f[b] and np.array_equal(a.x[0], r.x[b]), (mode, b)
        res[mode] = (r, solver, lam_b)
    (r1, solver, lam), (rb, sb, _), (r0, s0, lam0) = res["tail"], res["batched"], res["plain"]
    # the three paths reach the same optimum (a restart, or the cyclic reduction of the persistent kernel against the serial sweep, may send an
    # instance that sits on a fork between two local minima down the other branch -- both are KKT points, checked below; a handful in 16 384)
    for other in (rb, r0):
        same = np.abs(r1.f - other.f) <= 1e-8 * np.abs(other.f)
        assert same.mean() >= 0.9995, (same.mean(), np.flatnonzero(~same)[:10], r1.f[~same][:10], other.f[~same][:10])
    assert lam.shape == (B, 50, 14) and lam.min() >= 0.0 and np.abs(lam - lam0)[same].max() <= 1e-3 * max(1.0, lam.max())
    # the literal rows of the reference layout on a sample (k = velocity rows, a = linear rows, h = quaternion rows), reference-form KKT,
    # and the numpy port on the four former cap hitters
    orc = OracleRobot(KUKA_KIN)
    vl = np.asarray(orc.velocity_actuated_joint_limits)
    nlp = LimitedFigureEightNLP(orc, LINK, vlo=-vl, vup=vl, T=50)
    prob = StructuredFigureEight(orc, LINK, T=50)
    for n_b, b in enumerate(idx[:12]):
        x = r1.x[b]
        assert np.abs(nlp.a(x, qcs[b])).max() <= 1e-12 and np.abs(nlp.h(x, qcs[b])).max() <= 1e-9 and nlp.k(x, qcs[b]).min() >= -1e-8
        assert abs(nlp.f(x, qcs[b]) - r1.f[b]) <= 1e-9 * r1.f[b]
        k = kkt_reference_form(nlp, x, qcs[b], active_tol=1e-7)
        assert k["stationarity"] <= 1e-4 and k["feasibility"] <= 1e-8 and k["complementarity"] <= 1e-5, (b, k)
        if n_b < 4:
            s = solve_structured_lm(prob, qcs[b], max_iter=600, tol=1e-6, vlimits=(-vl, vl))
            assert s["status"] == 0 and abs(s["f"] - r1.f[b]) <= 1e-8 * s["f"], (b, s["status"], s["iters"], s["f"], r1.f[b])
    solver.backend.close()
    sb.backend.close()
    s0.backend.close()


def test_batches_beyond_the_plain_hand_over_start_in_the_persistent_kernel(hip_lib, monkeypatch):
    """Limit / velocity-limit handles run in k_tail_vel whatever the batch (the batched launches lose against it at every size measured):
    a batch above the plain family's hand-over threshold launches no batched iteration, converges everywhere, and an instance of it equals the same
    instance solved alone, bit for bit."""
    for k in ("OH_TAIL_VEL", "OH_TAIL_VEL_THRESHOLD", "OH_TAIL_THRESHOLD", "OH_COMPACTION"):
        oh_debug(monkeypatch, **{k: None})
    B = 24576
    rng = np.random.default_rng(9)
    qcs = QC0[None] + rng.uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx))
    x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    r = solver.solve_batch_arrays(x0, qcs)
    tm = solver.backend.timing()
    assert tm["iterations_launched"] == 0 and tm["tail_iterations"] > 0 and tm["compactions"] == 0, tm
    assert (r.status == 0).all() and (r.kkt[:, 0] <= 1e-6).all() and (r.kkt[:, 1] <= 1e-9).all(), (np.bincount(r.status, minlength=3), r.iters.max(), r.kkt[:, 0].max(), r.kkt[:, 1].max())
    vl = np.asarray(kuka.velocity_actuated_joint_limits)
    dQ = r.x[:, 350:].reshape(B, 49, 7)
    assert np.all(np.abs(dQ).max(1) <= vl + 1e-8)
    for b in (0, 12345, B - 1):
        a = solver.solve_batch_arrays(x0[b : b + 1], qcs[b : b + 1])
        assert a.status[0] == 0 and a.iters[0] == r.iters[b] and a.f[0] == r.f[b] and np.array_equal(a.x[0], r.x[b])


def test_horizon_beyond_the_persistent_kernel_compaction_moves_every_array_and_changes_nothing(hip_lib, monkeypatch):
    """T = 100 (beyond the persistent kernels' 64 knots): the batch drains through compactions.  Rounds 3 and 4 restarted the survivors at their
    accepted knots (round 3 retracted them again, which moved them by ~1e-10: of these 20 000 instances number 10 961 sat at the iteration cap
    inside the batch and converged in 133 steps alone; round 4 evaluated them as they were: same optimum for 19 996 of 20 000, bit-identical
    iterates for 2 897, because a restart rebuilds the stage data of the accepted point with that point's own multiplier estimates and drops a
    pending line search).  Round 5: an orientation-locked handle with inequality rows moves EVERY array of both slots with the instance
    (move_everything in oh_api.hip), nothing restarts -- pinned here: with and without compaction, and alone, every instance takes the same steps
    to the same bits.  The old restart compaction stays reachable (option compact_move_all = 0) and keeps its round-4 pin."""
    T, B = 100, 8192
    QC0 = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    rng = np.random.default_rng(T * 7 + 20000)
    qcs = (QC0[None] + rng.uniform(-0.1, 0.1, (20000, 7)))[10961 - B + 1:10961 + 1]  # the cap hitter of round 3 is the last of these
    x0 = np.repeat(qcs, T, axis=0).reshape(B, 7 * T)
    out = {}
    for mode in ("move", "none", "restart", "invariant", "invariant_none"):
        # ("invariant": the option batch_invariant on such a handle -- its own coarser schedule of the same moving compaction, last session of round 5; the option
        # also keeps the handle on the generic evaluation kernels, so its bits are compared with the option's own uncompacted run)
        oh_debug(monkeypatch, compaction="0" if mode == "none" else "1", compact_move_all="0" if mode == "restart" else "1",
                 batch_invariant="1" if mode.startswith("invariant") else None, invariant_compact_frac="0" if mode == "invariant_none" else None)
        kuka, solver = setup_solver(T=T, Tmax=10.0 * (T - 1) / 49.0, velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
        X0 = np.zeros((B, solver.opt.nx))
        X0[:, : 7 * T] = x0
        r = solver.solve_batch_arrays(X0, qcs)
        out[mode] = (np.array(r.status), np.array(r.f), solver.backend.timing()["compactions"], np.array(r.iters), np.array(r.x))
        if mode == "move":
            for b in (0, 4321, B - 1):
                a = solver.solve_batch_arrays(X0[b : b + 1], qcs[b : b + 1])
                assert a.status[0] == 0 and a.iters[0] == r.iters[b] and a.f[0] == r.f[b] and np.array_equal(a.x[0], r.x[b])
        solver.backend.close()
    assert out["move"][2] >= 5 and out["restart"][2] >= 5 and out["none"][2] == 0
    for mode in out:
        assert (out[mode][0] == 0).all(), mode
    assert np.array_equal(out["move"][3], out["none"][3]) and np.array_equal(out["move"][1], out["none"][1]) and np.array_equal(out["move"][4], out["none"][4])
    assert out["invariant"][2] >= 2 and out["invariant_none"][2] == 0 and all(np.array_equal(out["invariant"][k], out["invariant_none"][k]) for k in (1, 3, 4))
    assert (np.abs(out["invariant"][1] - out["none"][1]) <= 1e-9 * np.maximum(1.0, np.abs(out["none"][1]))).mean() >= 0.999
    rel = np.abs(out["restart"][1] - out["none"][1]) / np.maximum(1.0, np.abs(out["none"][1]))
    assert rel[-1] <= 1e-9  # instance 10 961
    assert (rel <= 1e-9).mean() >= 0.999
