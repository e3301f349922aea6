"""BASELINE config 3 (example/point_mass_mpc.py) on the CPU side: oracle restatement in the reference layout,
the reference-wired scipy SLSQP known answer of BASELINE.md section 5, the numpy port of the HIP interior-point kernel,
and the host mirror (builder counts of SURVEY 8(a) H3, lowering)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.pointmass_ipm import solve_pointmass_ipm
from oracle.problems import PointMassMPCNLP, point_mass_tick_parameters
from oracle.solvers import kkt_reference_form, scipy_minimize

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def test_oracle_counts_and_derivatives():
    nlp = PointMassMPCNLP()
    assert (nlp.nx, nlp.np_, nlp.nk, nlp.na, nlp.ng, nlp.nv) == (80, 84, 160, 42, 20, 264)
    p = point_mass_tick_parameters()
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, nlp.nx)
    g, Jg, H = nlp.df(x, p), nlp.dg(x, p), nlp.ddf(x, p)
    h = 1e-6
    for i in range(nlp.nx):
        d = np.zeros(nlp.nx)
        d[i] = h
        assert abs((nlp.f(x + d, p) - nlp.f(x - d, p)) / (2 * h) - g[i]) < 1e-7
        assert np.abs((nlp.g(x + d, p) - nlp.g(x - d, p)) / (2 * h) - Jg[:, i]).max() < 1e-8
        assert np.abs((nlp.df(x + d, p) - nlp.df(x - d, p)) / (2 * h) - H[:, i]).max() < 1e-6
    assert np.allclose(nlp.k(x, p), nlp.dk(x, p) @ x + nlp.k(np.zeros(nlp.nx), p))
    assert np.allclose(nlp.a(x, p), nlp.da(x, p) @ x + nlp.a(np.zeros(nlp.nx), p))
    a = nlp.a(x, p)
    assert np.allclose(nlp.v(x, p), np.concatenate([nlp.k(x, p), nlp.g(x, p), a, -a]))


def test_reference_wired_slsqp_known_answer():
    nlp = PointMassMPCNLP()
    p = point_mass_tick_parameters()
    r = scipy_minimize(nlp, np.zeros(nlp.nx), p, method="SLSQP", tol=1e-12, options={"maxiter": 500})
    assert r.success and abs(r.fun - 0.1759064919) < 1e-8
    assert abs(nlp.g(r.x, p).min()) < 1e-10  # the obstacle row is active
    k = kkt_reference_form(nlp, r.x, p)
    assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-10 and k["complementarity"] < 1e-8


def test_ipm_port_matches_golden():
    nlp = PointMassMPCNLP()
    d = np.load(os.path.join(GOLDEN, "pm_golden.npz"))
    for i in range(len(d["p"])):
        curr, dcurr, goal, obs = nlp.split_p(d["p"][i])
        r = solve_pointmass_ipm(20, 0.05, nlp.w, 1.5, 1.0, nlp.safe_sq, curr, dcurr, goal, obs, tol=1e-9)
        assert r["status"] == 0
        assert abs(r["f"] - d["f"][i]) <= 1e-7 * max(1.0, abs(d["f"][i]))
        x = np.concatenate([r["Y"].T.reshape(-1), r["V"].T.reshape(-1)])
        assert np.abs(nlp.a(x, d["p"][i])).max() < 1e-14
        k = kkt_reference_form(nlp, x, d["p"][i])
        assert k["stationarity"] < 1e-5 and k["feasibility"] < 1e-9 and k["complementarity"] < 1e-8


def test_builder_counts_and_lowering():
    from examples.point_mass_mpc import Controller
    from optas_amd import _lib
    from optas_amd.lowering import lower
    from optas_amd.optimization import QuadraticCostNonlinearConstraints

    c = Controller(build_only=True)
    o = c.optimization
    assert isinstance(o, QuadraticCostNonlinearConstraints)
    assert (o.nx, o.np, o.nk, o.na, o.ng, o.nh, o.nv) == (80, 84, 160, 42, 20, 0, 264)
    assert list(o.decision_variables.keys()) == ["point_mass/y/x", "point_mass/dy/x"]
    assert list(o.parameters.keys()) == ["curr", "dcurr", "goal", "obs"]
    assert list(o.lin_ineq_constraints.keys()) == [
        "__point_mass_model_limit_0___l", "__point_mass_model_limit_0___r", "__point_mass_model_limit_1___l", "__point_mass_model_limit_1___r"]
    assert list(o.ineq_constraints.keys()) == [f"obs_avoid_{i}" for i in range(20)]
    kind, spec = lower(o)
    assert kind == _lib.OH_PROBLEM_POINT_MASS_MPC
    assert (spec.T, spec.dt, spec.ylim, spec.vlim) == (20, 0.05, 1.5, 1.0) and abs(spec.safe - 0.3) < 1e-15 and abs(spec.w_acc - 0.0025 / 20) < 1e-18
    # parameter vector layout = the oracle's
    p = o.parameters.dict2vec({"curr": [0.1, 0.2], "dcurr": [0.3, 0.4], "goal": np.arange(40.0).reshape(2, 20), "obs": -np.arange(40.0).reshape(2, 20)})
    assert np.array_equal(p, PointMassMPCNLP.pack_p([0.1, 0.2], [0.3, 0.4], np.arange(40.0).reshape(2, 20), -np.arange(40.0).reshape(2, 20)))


def test_diagnostics_match_oracle_rows():
    """Solver.evaluate_cost / violated_constraints (solver.py:167-237, 269-314) on the mirror builder reproduce the
    oracle's f, k, g, a in the reference's row order (no FK in this problem, so this runs without a GPU)."""
    from examples.point_mass_mpc import Controller
    from optas_amd.solver import Solver

    class Probe(Solver):
        def setup(self):
            return self

        def _solve(self):
            return None

        def stats(self):
            return None

        def did_solve(self):
            return True

        def number_of_iterations(self):
            return 0

    o = Controller(build_only=True).optimization
    s = Probe(o)
    nlp = PointMassMPCNLP()
    p = point_mass_tick_parameters()
    x = np.random.default_rng(1).uniform(-1, 1, 80)
    xd, pd = o.decision_variables.vec2dict(x), o.parameters.vec2dict(p)
    assert abs(s.evaluate_cost(xd, pd) - nlp.f(x, p)) < 1e-13
    assert len(s.evaluate_cost_terms(xd, pd)) == 2
    lin_eq, eq, lin_ineq, ineq = s.violated_constraints(xd, pd)
    cat = lambda lst: np.concatenate([v.diff.T.reshape(-1) for v in lst])
    assert np.abs(cat(lin_ineq) - nlp.k(x, p)).max() < 1e-14 and np.abs(cat(ineq) - nlp.g(x, p)).max() < 1e-14
    assert np.abs(cat(lin_eq) - nlp.a(x, p)).max() < 1e-14 and eq == []
    assert [v.label for v in ineq] == [f"obs_avoid_{i}" for i in range(20)] and ineq[0].ctype == "ineq"
    assert np.array_equal(lin_ineq[0].pattern, lin_ineq[0].diff >= 0.0)


def test_planner_variant_builder_lowering_and_port():
    """example/point_mass_planner.py: builder counts, the mirrored rows against the literal restatement, lowering to the point-mass
    family's planner options, and the numpy port against scipy SLSQP (reference wiring) and the reference-form KKT."""
    import os
    import sys

    from scipy.optimize import minimize

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from examples.point_mass_planner import Planner
    from optas_amd import _lib
    from optas_amd.lowering import lower
    from oracle.pointmass_ipm import solve_pointmass_ipm
    from oracle.problems import PointMassPlannerNLP
    from oracle.solvers import kkt_reference_form, scipy_minimize

    o = Planner(build_only=True).optimization
    nlp = PointMassPlannerNLP()
    assert (o.nx, o.np, o.nk, o.na, o.ng, o.nh, o.nv) == (nlp.nx, nlp.np_, nlp.nk, nlp.na, nlp.ng, 0, nlp.nv) == (180, 4, 360, 94, 45, 0, 593)
    rng = np.random.default_rng(3)
    x, p = rng.normal(size=180), np.array([-1.0, -0.8, 1.0, 0.9])
    assert abs(o.f(x, p) - nlp.f(x, p)) < 1e-12 and np.abs(o.v(x, p) - nlp.v(x, p)).max() < 1e-13
    kind, spec = lower(o)
    assert kind == _lib.OH_PROBLEM_POINT_MASS_MPC and spec.planner is not None and abs(spec.planner["w_vel"] - 0.01 / 45) < 1e-18
    assert (spec.T, spec.dt, spec.ylim, spec.vlim) == (45, 0.1, 1.5, 1.0) and abs(spec.w_acc - 0.005 / 45) < 1e-18 and abs(spec.safe - 0.3) < 1e-15
    T = 45
    G, O = np.tile(p[2:4][:, None], (1, T)), np.zeros((2, T))
    r = solve_pointmass_ipm(T, 0.1, nlp.w, 1.5, 1.0, nlp.safe_sq, p[:2], np.zeros(2), G, O, tol=1e-9, max_iter=200, track_final_only=True,
                            w_vel=nlp.w_vel, fix_final_velocity=True)
    assert r["status"] == 0 and np.abs(r["V"][:, -1]).max() < 1e-12
    xs = np.concatenate([r["Y"].T.reshape(-1), r["V"].T.reshape(-1)])
    assert abs(nlp.f(xs, p) - r["f"]) < 1e-12 and np.abs(nlp.a(xs, p)).max() < 1e-12 and nlp.g(xs, p).min() > -1e-9 and nlp.k(xs, p).min() > -1e-9
    k = kkt_reference_form(nlp, xs, p, active_tol=1e-3)
    assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-9
    s = scipy_minimize(nlp, xs + 0.0, p, method="SLSQP", tol=1e-12, options={"maxiter": 300})  # polish from the port's answer: no lower point nearby
    assert s.success and s.fun >= r["f"] - 1e-9 and abs(s.fun - r["f"]) < 1e-8
