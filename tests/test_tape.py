"""Generic tape family (OH_PROBLEM_TAPE): the host compiles a problem's expression trees into one scalar instruction tape
(optas_amd/tape.py, the counterpart of the CasADi SX tape of the reference) and the GPU interprets it.  CPU: the compiled tapes against the
oracle's literal NLPs (values, gradients, Jacobian rows: independent implementations) and the numpy port of the solver against scipy.
GPU: the kernel against the port and the golden optima."""
import os
import sys

import numpy as np
import pytest
from scipy.optimize import minimize

from conftest import GOLDEN, KUKA_KIN, SEED
from optas_amd.tape import compile_problem
from oracle import tape_ref
from oracle.problems import DualArmNLP, IKExampleNLP, PointMassMPCNLP
from oracle.robot import OracleRobot

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _check_tape(o, nlp, x, p, rng):
    tp = compile_problem(o)
    v = tape_ref.forward(tp, x, p)
    assert abs(v[tp.out_cost] - nlp.f(x, p)) < 1e-12 * max(1.0, abs(nlp.f(x, p)))
    ref_rows = np.concatenate([nlp.k(x, p), nlp.g(x, p), nlp.a(x, p), nlp.h(x, p)])
    assert (tp.n_ineq, tp.n_eq) == (nlp.nk + nlp.ng, nlp.na + nlp.nh) and np.abs(v[tp.out_rows] - ref_rows).max() < 1e-12
    assert np.abs(tape_ref.reverse(tp, v, {tp.out_cost: 1.0}) - nlp.df(x, p)).max() < 1e-11
    J = np.vstack([nlp.dk(x, p), nlp.dg(x, p), nlp.da(x, p), nlp.dh(x, p)])
    for r in rng.choice(len(tp.out_rows), min(6, len(tp.out_rows)), replace=False):
        assert np.abs(tape_ref.reverse(tp, v, {int(tp.out_rows[r]): 1.0}) - J[r]).max() < 1e-11
    return tp


def test_compiled_tapes_match_the_literal_restatements():
    from examples.dual_arm import setup_solver as dual
    from examples.example import setup_solver as ik
    from examples.point_mass_mpc import Controller

    rng = np.random.default_rng(SEED)
    o = Controller(build_only=True).optimization
    tp = _check_tape(o, PointMassMPCNLP(), rng.normal(size=o.nx), rng.normal(size=o.np), rng)
    assert len(tp.op) < 1000
    _, o = ik(build_only=True)
    tp = _check_tape(o, IKExampleNLP(OracleRobot(KUKA_KIN), "end_effector_ball"), rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 10), rng)
    assert len(tp.op) < 400  # the whole 7-joint chain with its constants folded and shared
    _, o = dual(T=5, build_only=True)
    rl = OracleRobot(KUKA_KIN, name="kukal")
    rl.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
    rr = OracleRobot(KUKA_KIN, name="kukar")
    rr.add_base_frame("global_world", xyz=[0.0, 0.25, 0.0])
    _check_tape(o, DualArmNLP(rl, rr, T=5), rng.uniform(-1, 1, o.nx), rng.uniform(-1, 1, o.np), rng)


def test_port_solves_ik_and_planar_ik():
    from examples.example import setup_solver as ik
    from examples.planar_ik import setup_solver as planar
    from optas_amd import _lib
    from optas_amd.lowering import lower

    g = np.load(os.path.join(GOLDEN, "ik_golden.npz"))
    _, o = ik(build_only=True)
    tp = compile_problem(o)
    for i in (0, 5, 17):
        r = tape_ref.solve_tape_al(tp, g["x0"][i], g["p"][i])
        assert r["status"] == 0 and abs(r["f"] - g["f"][i]) < 1e-7 and np.abs(r["x"] - g["x"][i]).max() < 1e-4 and r["feas"] < 1e-9
    _, o = planar(build_only=True)
    assert (o.nx, o.np, o.nk, o.ng, o.nh) == (3, 0, 6, 2, 2)
    kind, spec = lower(o)
    assert kind == _lib.OH_PROBLEM_TAPE  # no hand-written family takes the heading row
    tp = spec.tape
    x0, p = np.array([np.pi / 2, 0.0, 0.0]), np.zeros(0)
    r = tape_ref.solve_tape_al(tp, x0, p)
    assert r["status"] == 0 and r["feas"] < 1e-9
    fun = lambda x: (lambda v: (v[tp.out_cost], tape_ref.reverse(tp, v, {tp.out_cost: 1.0})))(tape_ref.forward(tp, x, p))
    cons = [{"type": "ineq", "fun": lambda x: tape_ref.forward(tp, x, p)[tp.out_rows[: tp.n_ineq]]},
            {"type": "eq", "fun": lambda x: tape_ref.forward(tp, x, p)[tp.out_rows[tp.n_ineq :]]}]
    s = minimize(fun, r["x"], jac=True, method="SLSQP", constraints=cons, tol=1e-12, options={"maxiter": 200})
    assert s.success and abs(s.fun - r["f"]) < 1e-6 and np.abs(s.x - r["x"]).max() < 1e-4  # a local optimum: SLSQP cannot improve on it
    v = tape_ref.forward(tp, r["x"], p)
    assert abs(np.sum(r["x"]) + 70.0 * np.pi / 180.0) < 1e-8  # the heading row (sum of the planar joint angles >= -70 deg) is active


def test_generated_kernel_source_compiles_for_gfx950():
    """oh_tape_compile: code generation + hiprtc need no device, so the build of the jit path is checked here."""
    from examples.planar_ik import setup_solver as planar
    from optas_amd.backend import TapeBackend

    tp = compile_problem(planar(build_only=True)[1])
    src, size = TapeBackend.generated_source(tp)
    assert size > 10000 and "k_tape_jit" in src and "tape_solve_instance" in src  # the solver text of oh_tape_solver.h rides in front
    body = src[src.index("struct JitEval") :]
    assert body.count("sin(") == 6 and body.count("cos(") == 6  # 3 joints: forward value and the partial of the other one; hipcc shares them
    assert "#pragma clang fp contract(off)" in body  # same IEEE operations as the interpreter and the numpy port
    keep = []
    desc = TapeBackend.descriptor(tp, keep)
    keep[0][5] = 99  # unknown opcode (13 .. 24 joined the vocabulary in round 4)
    from optas_amd import _lib
    import ctypes as C

    assert _lib.load().oh_tape_compile(C.byref(desc), None, None, 0, None) == _lib.OH_ERR_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [True, False])
def test_tape_kernel_matches_port_and_goldens(hip_lib, jit):
    from examples.example import setup_solver as ik
    from examples.planar_ik import setup_solver as planar
    from optas_amd.backend import TapeBackend

    g = np.load(os.path.join(GOLDEN, "ik_golden.npz"))
    _, o = ik(build_only=True)
    tp = compile_problem(o)
    be = TapeBackend(tp, jit=jit)
    res = be.solve(g["x0"], g["p"])
    if jit:  # generated code and interpreter perform the same operations in the same order
        ref = TapeBackend(tp, jit=False).solve(g["x0"], g["p"])
        assert np.array_equal(res.x, ref.x) and np.array_equal(res.iters, ref.iters) and np.array_equal(res.f, ref.f)
    lam, mu = be.multipliers(len(g["p"]))
    assert (res.status == 0).all() and np.abs(res.f - g["f"]).max() < 1e-7 and np.abs(res.x - g["x"]).max() < 1e-4 and res.kkt[:, 1].max() < 1e-9
    for i in (0, 5, 17):
        r = tape_ref.solve_tape_al(tp, g["x0"][i], g["p"][i])
        assert abs(int(res.iters[i]) - r["evals"]) <= 2 and np.abs(res.x[i] - r["x"]).max() < 1e-7
        assert np.abs(lam[i] - r["lam"]).max() < 1e-4 and np.abs(mu[i] - r["mu"]).max() < 1e-4
    assert int((lam > 0).sum(axis=1).max()) >= 1 and (lam >= 0).all()  # some instances sit on joint limits
    # planar_ik.py through the Solver interface
    robot, solver = planar(solver_options={"jit": jit})
    solver.reset_initial_seed({f"{robot.get_name()}/q/x": [np.pi / 2.0, 0.0, 0.0]})
    sol = solver.solve()
    q = np.asarray(sol[f"{robot.get_name()}/q"]).reshape(-1)
    r = tape_ref.solve_tape_al(solver._spec.tape, np.array([np.pi / 2, 0.0, 0.0]), np.zeros(0))
    assert solver.did_solve() and np.abs(q - r["x"]).max() < 1e-7 and abs(solver.stats()["f"][0] - r["f"]) < 1e-9
    assert np.abs(np.asarray(robot.get_global_link_position("end", q)).reshape(-1)[:2] - [1.2, 0.2]).max() < 1e-8  # the FK row, on the GPU kinematics
    o = solver.opt
    x = o.decision_variables.dict2vec(sol)
    assert o.k(x, np.zeros(0)).min() > -1e-9 and o.g(x, np.zeros(0)).min() > -1e-9 and np.abs(o.h(x, np.zeros(0))).max() < 1e-8
