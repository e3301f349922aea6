"""Numeric evaluation of ``optas_amd.expr`` trees at a given (x, p): the diagnostics of the reference's Solver
(``evaluate_cost``, ``evaluate_cost_terms``, ``violated_constraints``, optas/solver.py:167-237,269-314) need the value
of individual cost terms / constraint blocks.  Forward kinematics inside a tree goes through liboptas_hip
(``RobotModel.get_global_link_*`` -> ``oh_fk_jac``); everything else is elementwise host bookkeeping on the results.
"""
from __future__ import annotations

import numpy as np

from .builder import IntegrationResidual
from .expr import Add, Atan2, Block, Const, Expr, LinkFunction, MatMul, Mul, ParamCol, ParamRef, PathInFrame, RneaFunction, RobotStates, Rows, Scale, Square, StateCols, StateRef, Sub, SumSqr, VCat, VarRef


def _block(container, vec, label):
    off = container.offsets()[label]
    m, n = container[label].shape
    return vec[off : off + m * n].reshape(n, m).T


def evaluate(e: Expr, opt, x: np.ndarray, p: np.ndarray) -> np.ndarray:
    """Value of node ``e`` as a 2-D array (rows x cols like the CasADi matrix it stands for).  Link functions of parameters only
    (``J(qc)``, ``p(qc)``) are memoised per parameter vector: reading P, q, M, c off a QP evaluates the trees dozens of times with
    the same p, and each of those nodes is a round trip to the GPU."""
    if isinstance(e, Const):
        return e.value
    if isinstance(e, LinkFunction) and e.q.degree() == 0:
        cache = getattr(opt, "_link_cache", None)
        key = p.tobytes()
        if cache is None or cache[0] != key:
            cache = (key, {})
            opt._link_cache = cache
        if id(e) not in cache[1]:
            cache[1][id(e)] = _evaluate(e, opt, x, p)
        return cache[1][id(e)]
    return _evaluate(e, opt, x, p)


def _evaluate(e: Expr, opt, x: np.ndarray, p: np.ndarray) -> np.ndarray:
    if isinstance(e, Const):
        return e.value
    if isinstance(e, ParamRef):
        return _block(opt.parameters, p, e.name)
    if isinstance(e, ParamCol):
        return _block(opt.parameters, p, e.param.name)[:, [e.col]]
    if isinstance(e, StateRef):
        full = _block(opt.decision_variables, x, e.var_name)
        return full if e.t is None else full[:, [e.t]]
    if isinstance(e, StateCols):
        return _block(opt.decision_variables, x, e.state.var_name)[:, e.lo : e.hi]
    if isinstance(e, RobotStates):
        X = evaluate(e.states, opt, x, p)
        Pm = evaluate(e.params, opt, x, p)
        full = np.zeros(e.shape)
        full[list(e.opt_idx), :] = X
        full[list(e.par_idx), :] = Pm
        return full
    if isinstance(e, Rows):
        return evaluate(e.a, opt, x, p)[list(e.idx), :]
    if isinstance(e, Block):
        return np.asarray(evaluate(e.a, opt, x, p))[np.ix_(list(e.ridx), list(e.cidx))]
    if isinstance(e, VarRef):
        return _block(opt.decision_variables, x, e.var_name)
    if isinstance(e, LinkFunction):
        q = evaluate(e.q, opt, x, p)
        if e.what == "position":
            return np.asarray(e.robot.get_global_link_position(e.link, q)).reshape(3, -1)
        if e.what == "quaternion":
            return np.asarray(e.robot.get_global_link_quaternion(e.link, q)).reshape(4, -1)
        if e.what == "geometric_jacobian":
            return np.asarray(e.robot.get_global_link_geometric_jacobian(e.link, q.reshape(-1)))
        return np.asarray(e.robot.get_global_link_rotation(e.link, q.reshape(-1)))
    if isinstance(e, RneaFunction):
        return np.asarray(e.robot.rnea(evaluate(e.q, opt, x, p), evaluate(e.qd, opt, x, p), evaluate(e.qdd, opt, x, p))).reshape(e.shape)
    if isinstance(e, PathInFrame):
        return evaluate(e.origin, opt, x, p).reshape(3, 1) + evaluate(e.rotation, opt, x, p) @ e.local
    if isinstance(e, IntegrationResidual):
        X = evaluate(e.x, opt, x, p)
        Xd = evaluate(e.xd, opt, x, p)[:, : e.n]
        return X[:, :-1][:, : e.n] + e.dt[None, :] * Xd - X[:, 1:][:, : e.n]
    if isinstance(e, Sub):
        return evaluate(e.a, opt, x, p) - evaluate(e.b, opt, x, p)
    if isinstance(e, Add):
        return evaluate(e.a, opt, x, p) + evaluate(e.b, opt, x, p)
    if isinstance(e, Scale):
        return e.w * evaluate(e.a, opt, x, p)
    if isinstance(e, Atan2):
        return np.arctan2(evaluate(e.y, opt, x, p), evaluate(e.x, opt, x, p))
    if isinstance(e, MatMul):
        return evaluate(e.a, opt, x, p) @ evaluate(e.b, opt, x, p)
    if isinstance(e, VCat):
        return np.vstack([np.broadcast_to(evaluate(q_, opt, x, p), q_.shape) for q_ in e.parts])
    if isinstance(e, Mul):
        return evaluate(e.a, opt, x, p) * evaluate(e.b, opt, x, p)
    if isinstance(e, Square):
        v = evaluate(e.a, opt, x, p)
        return v * v
    if isinstance(e, SumSqr):
        v = evaluate(e.a, opt, x, p)
        return np.array([[float(np.sum(v * v))]])
    raise NotImplementedError(f"cannot evaluate {type(e).__name__}")
