"""The oracle's NLP restatements and solvers: reference row counts, analytic
derivatives vs finite differences, the reference's only numeric solver pin (Booth -> (1,3),
tests/test_solver.py:46-54), and the committed golden solutions."""
import numpy as np
import pytest

from conftest import KUKA_KIN
from oracle.problems import BoothNLP, FigureEightNLP, IKExampleNLP
from oracle.robot import OracleRobot
from oracle.solvers import dense_sqp, kkt_reference_form, scipy_minimize
from oracle.structured import StructuredFigureEight, solve_structured, solve_structured_lm

LINK = "end_effector_ball"


@pytest.fixture(scope="module")
def kuka():
    return OracleRobot(KUKA_KIN)


@pytest.mark.parametrize("method", ["SLSQP", "BFGS", "CG", "L-BFGS-B", "TNC", "Newton-CG", "trust-constr"])
def test_booth_known_answer(method):
    # reference: every backend must return x=1, y=3 for (a,b)=(2,7) from seed (0,0)
    r = scipy_minimize(BoothNLP(), [0.0, 0.0], np.array([2.0, 7.0]), method=method, tol=1e-6)
    assert np.isclose(r.x, [1.0, 3.0]).all()


def test_ik_example_counts_and_solution(kuka, golden_nlp):
    ik = IKExampleNLP(kuka, LINK)
    assert (ik.nx, ik.np_, ik.nk, ik.nh, ik.nv) == (7, 10, 14, 3, 20)
    p = golden_nlp["ik_p"]
    r = scipy_minimize(ik, np.zeros(7), p, method="SLSQP", tol=1e-12, options={"maxiter": 500})
    assert r.success
    assert abs(r.fun - 0.29579887518) < 1e-9
    assert abs(r.fun - float(golden_nlp["ik_f"])) < 1e-12
    k = kkt_reference_form(ik, r.x, p)
    assert k["stationarity"] < 1e-6 and k["feasibility"] < 1e-10 and k["complementarity"] < 1e-8
    # same optimum from the nominal seed
    r2 = scipy_minimize(ik, p[:7], p, method="SLSQP", tol=1e-12)
    assert abs(r2.fun - r.fun) < 1e-8


def test_figure_eight_counts_layout(kuka):
    nlp = FigureEightNLP(kuka, LINK, T=50)
    assert (nlp.nx, nlp.np_, nlp.na, nlp.nh, nlp.nv) == (693, 7, 357, 200, 1114)
    assert np.isclose(nlp.dt, 10.0 / 49.0)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    x0 = nlp.seed(qc)
    Q, dQ = nlp.split(x0)
    assert Q.shape == (7, 50) and dQ.shape == (7, 49) and np.allclose(Q, qc[:, None]) and not dQ.any()
    assert x0[7 * 3 + 2] == qc[2]  # x[7t + j] = q_j(t)
    assert np.abs(nlp.a(x0, qc)).max() == 0.0 and np.abs(nlp.h(x0, qc)).max() < 1e-15
    assert abs(nlp.f(x0, qc) - 1225.0) < 1e-9
    v = nlp.v(x0, qc)
    assert v.shape == (1114,)
    a, h = nlp.a(x0 + 0.01, qc), nlp.h(x0 + 0.01, qc)
    assert np.allclose(nlp.v(x0 + 0.01, qc), np.concatenate([a, -a, h, -h]))


def test_figure_eight_derivatives_fd(kuka):
    nlp = FigureEightNLP(kuka, LINK, T=6, Tmax=1.0)
    rng = np.random.default_rng(5)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    x = nlp.seed(qc) + 0.05 * rng.standard_normal(nlp.nx)
    g, Jh, Ja = nlp.df(x, qc), nlp.dh(x, qc), nlp.da(x, qc)
    lam = rng.standard_normal(nlp.nh)
    H = nlp.hess_lagrangian(x, qc, lam)
    h = 1e-6
    for i in range(nlp.nx):
        d = np.zeros(nlp.nx)
        d[i] = h
        assert abs((nlp.f(x + d, qc) - nlp.f(x - d, qc)) / (2 * h) - g[i]) < 1e-5
        assert np.abs((nlp.h(x + d, qc) - nlp.h(x - d, qc)) / (2 * h) - Jh[:, i]).max() < 1e-8
        assert np.abs((nlp.a(x + d, qc) - nlp.a(x - d, qc)) / (2 * h) - Ja[:, i]).max() < 1e-8
        gl = lambda xx: nlp.df(xx, qc) + nlp.dh(xx, qc).T @ lam
        assert np.abs((gl(x + d) - gl(x - d)) / (2 * h) - H[:, i]).max() < 1e-4


def test_figure_eight_golden_is_kkt_point(kuka, golden_nlp):
    nlp = FigureEightNLP(kuka, LINK, T=50)
    x, qc = golden_nlp["fig8_x"], golden_nlp["fig8_qc"]
    assert abs(nlp.f(x, qc) - 8.498170214656) < 1e-10  # SURVEY App. D optimum
    k = kkt_reference_form(nlp, x, qc)
    assert k["stationarity"] < 1e-9 and k["feasibility"] < 1e-12 and k["complementarity"] < 1e-9
    for i in range(len(golden_nlp["fig8_pert_qc"])):
        k = kkt_reference_form(nlp, golden_nlp["fig8_pert_x"][i], golden_nlp["fig8_pert_qc"][i])
        assert k["stationarity"] < 1e-8 and k["feasibility"] < 1e-9  # retraction tolerance 1e-10 in the port


def test_structured_and_dense_oracles_agree(kuka, golden_nlp):
    # two independent CPU algorithms (banded null-space GN vs dense SVD Newton) reach the same optimum
    qc = golden_nlp["fig8_qc"]
    for T in (5, 12):
        Tmax = 10.0 * (T - 1) / 49.0
        prob = StructuredFigureEight(kuka, LINK, T=T, Tmax=Tmax)
        s = solve_structured(prob, qc, max_iter=200, tol=1e-10, exact=False)
        assert abs(s["f"] - float(golden_nlp[f"fig8_T{T}_f"])) < 1e-9
    nl = FigureEightNLP(kuka, LINK, T=5, Tmax=10.0 * 4 / 49.0)
    d = dense_sqp(nl, nl.seed(qc), qc, tol=1e-11)
    assert d["converged"] and abs(d["f"] - float(golden_nlp["fig8_T5_f"])) < 1e-10
    # scipy SLSQP in the reference wiring (v >= 0 with the literal 4-row quaternion rows) on the small case
    r = scipy_minimize(nl, nl.seed(qc), qc, method="SLSQP", tol=1e-12, options={"maxiter": 300})
    assert nl.f(r.x, qc) >= d["f"] - 1e-6  # SLSQP may stall on the rank-deficient rows, never beats the KKT point
    s50 = solve_structured(StructuredFigureEight(kuka, LINK, T=50), qc, max_iter=300, tol=1e-9, exact=False)
    assert abs(s50["f"] - float(golden_nlp["fig8_f"])) < 1e-9
    # the port of the HIP state machine (retraction + LM ratio test) reaches the same optimum
    lm = solve_structured_lm(StructuredFigureEight(kuka, LINK, T=50), qc, max_iter=300, tol=1e-7)
    assert lm["status"] == 0 and abs(lm["f"] - float(golden_nlp["fig8_f"])) < 1e-9 and lm["feas"] < 1e-9
