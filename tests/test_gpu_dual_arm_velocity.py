"""Joint-velocity limits on the position-tracking family (round-2 verdict, Missing 3): example/dual_arm.py plus, per arm,
``builder.enforce_model_limits(name, time_deriv=1)`` (builder.py:471-509).  The rows dq_t - vlo >= 0, vup - dq_t >= 0 on dq_t = (q_{t+1} - q_t)/dt
are lowered to the guarded kernels of csrc/oh_free.hip (k_couple_free_vel; both sweeps carry the diagonal coupling blocks).  Checked against the
numpy port (oracle/guarded.py:solve_free_al(vlimits=...)), the reference-form KKT conditions on the literal layout
(oracle/problems.py:GuardedDualArmNLP(vlimits=...), 2 x 686 extra k rows), with and without the other inequality rows, and for both sweeps."""
import os
import sys

import numpy as np
import pytest

from conftest import KUKA_KIN, SEED, oh_debug
from oracle.guarded import Guards, solve_free_al
from oracle.problems import GuardedDualArmNLP, dual_arm_offsets
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import FoldedChain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.dual_arm import N_OBSTACLES, SPHERE_LINKS, obstacle_parameters, setup_solver  # noqa: E402

pytestmark = pytest.mark.gpu
QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
VMAX = 0.06  # rad/s: the unconstrained optimum of dual_arm.py runs joint 3 at 0.099 and joint 1 at 0.074


def _robots():
    rl = OracleRobot(KUKA_KIN, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(KUKA_KIN, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    return rl, rr


@pytest.mark.parametrize("others", [False, True])
def test_dual_arm_with_velocity_limits(hip_lib, others):
    T = 50
    vl = np.full(7, VMAX)
    (kl, kr), solver = setup_solver(T=T, limits=others, collision=others, velocity_limits=(-vl, vl), solver_options={"max_iter": 600, "tol": 1e-7})
    o = solver.opt
    assert o.nk == (4 * 7 * T if others else 0) + 4 * 7 * (T - 1)
    pd = {"qcl": QC, "qcr": QC + 0.02}
    if others:
        pd.update(obstacle_parameters())
    solver.reset_parameters(pd)
    solver.reset_initial_seed({"kukal/q/x": np.tile(pd["qcl"].reshape(-1, 1), (1, T)), "kukar/q/x": np.tile(pd["qcr"].reshape(-1, 1), (1, T))})
    sol = solver.solve()
    st = solver.stats()
    assert solver.did_solve(), st
    dQl, dQr = np.asarray(sol["kukal/dq"]), np.asarray(sol["kukar/dq"])
    assert max(np.abs(dQl).max(), np.abs(dQr).max()) <= VMAX + 1e-8 and np.abs(dQl).max() >= VMAX - 1e-6  # the rows bind
    rl, rr = _robots()
    nlp = GuardedDualArmNLP(rl, rr, SPHERE_LINKS if others else [], N_OBSTACLES if others else 0, T=T, limits=others, vlimits=(-vl, vl))
    x = o.decision_variables.dict2vec(sol)
    p = o.parameters.dict2vec(pd)
    assert (nlp.nx, nlp.nk, nlp.na) == (o.nx, o.nk, o.na)
    rng = np.random.default_rng(SEED)
    xr = rng.uniform(-1, 1, o.nx)
    assert np.abs(o.k(xr, p) - nlp.k(xr, p)).max() <= 1e-14 and np.abs(o.dk(xr, p) - nlp.dk(xr, p)).max() == 0.0  # same rows, same order
    assert abs(nlp.f(x, p) - st["f"][0]) <= 1e-12 and np.abs(nlp.a(x, p)).max() <= 1e-13 and nlp.k(x, p).min() >= -1e-8
    k = kkt_reference_form(nlp, x, p, active_tol=1e-6)
    assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-8 and k["complementarity"] <= 1e-6, k
    # the numpy port, arm by arm: same optimum, same active velocity rows, multipliers
    off = dual_arm_offsets(T)
    f_port = 0.0
    for (a, be), rob, arm, qc in zip(solver.backend.arms, (rl, rr), ("l", "r"), (pd["qcl"], pd["qcr"])):
        ch = FoldedChain(rob, "end_effector_ball")
        G = Guards(lo=rob.lower_actuated_joint_limits, up=rob.upper_actuated_joint_limits, links=SPHERE_LINKS, link_radii=np.full(4, 0.15),
                   obs_pos=np.array([[0.55, 0.0, 0.1 * (i + 1)] for i in range(N_OBSTACLES)]), obs_radii=np.full(N_OBSTACLES, 0.1)) if others else Guards()
        s = solve_free_al(ch, T, 10.0 / (T - 1), off[arm].T, qc, G, Q0=np.tile(qc, (T, 1)), rho0=10.0, exact=False, vlimits=(-vl, vl), max_iter=600, tol=1e-7)
        assert s["status"] == 0
        f_port += s["f"]
        Qg = np.asarray(sol[f"kuka{arm}/q"]).T
        assert np.abs(Qg - s["Q"]).max() <= 2e-4  # ~1e-5 rad of play along the weakly curved directions, more where velocity rows pin the path
        lam = be.multipliers(1)[0]  # (T, NC + 14): the velocity rows of knot t are those of dq_t, in the reference's order [dq_t - vlo; vup - dq_t]
        lv = np.zeros((T, 14))
        lv[: T - 1] = s["lam_v"]
        assert lam.shape[1] == (14 + len(SPHERE_LINKS) * N_OBSTACLES if others else 0) + 14
        assert np.abs(lam[:, -14:] - lv).max() <= 1e-3 * max(1.0, lv.max()) and ((lam[:, -14:] > 1e-9) == (lv > 1e-9)).mean() >= 0.99
    assert abs(f_port - st["f"][0]) <= 1e-8 * max(1.0, f_port)


def test_velocity_limited_arms_batch_both_sweeps_and_compaction(hip_lib, monkeypatch):
    """A batch large enough to be compacted while it drains, solved by the serial sweep and by cyclic reduction: same optima, multipliers move
    with the instances."""
    from optas_amd.backend import MultiArmBackend
    from optas_amd.lowering import MultiArmSpec, lower

    T, B = 50, 1024
    vl = np.full(7, VMAX)
    (kl, kr), o = setup_solver(T=T, build_only=True, velocity_limits=(-vl, vl))
    kind, spec = lower(o)
    assert isinstance(spec, MultiArmSpec) and spec.arms[0].guards.vlo is not None
    rng = np.random.default_rng(SEED + 7)
    qcl, qcr = QC + rng.uniform(-0.1, 0.1, (B, 7)), QC + rng.uniform(-0.1, 0.1, (B, 7))
    P = np.concatenate([qcl, qcr], 1)
    X0 = np.zeros((B, o.nx))
    xoff = o.decision_variables.offsets()
    for name, qc in (("kukal/q/x", qcl), ("kukar/q/x", qcr)):
        X0[:, xoff[name] : xoff[name] + 7 * T] = np.tile(qc, (1, T))
    out = {}
    for mode in ("0", "4096"):
        oh_debug(monkeypatch, free_pcr_max=mode)
        mb = MultiArmBackend(spec, o, max_iter=600)
        r = mb.solve(X0, P)
        out[mode] = (r, [be.multipliers(B) for _, be in mb.arms], [be.timing()["compactions"] for _, be in mb.arms])
        mb.close()
    (r0, l0, c0), (r1, l1, c1) = out["0"], out["4096"]
    assert (r0.status == 0).mean() >= 0.999 and (r1.status == 0).mean() >= 0.999, ((r0.status == 0).mean(), (r1.status == 0).mean())
    assert min(c0) >= 1  # the batch was compacted
    ok = (r0.status == 0) & (r1.status == 0)
    # (two elimination orders, each stopped at a reduced gradient of 1e-6 with multipliers converged to 1e-9: the objectives agree to ~1e-7 relative)
    assert np.abs(r0.f - r1.f)[ok].max() <= 1e-6 * np.abs(r0.f).max()
    dq = r0.x[:, xoff["kukal/dq/x"] : xoff["kukal/dq/x"] + 7 * (T - 1)]
    assert np.abs(dq[ok]).max() <= VMAX + 1e-8 and (np.abs(dq[ok]).max(1) >= VMAX - 1e-6).mean() > 0.5
    for a, b in zip(l0, l1):
        assert a.shape == (B, T, 14) and a.min() >= 0.0 and np.abs(a[ok] - b[ok]).max() <= 1e-3 * max(1.0, np.abs(a).max())
