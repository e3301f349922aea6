"""``Solver`` interface of the reference (optas/solver.py:61-314) and the MI355X backend that drops in
beside ``CasADiSolver`` / ``ScipyMinimizeSolver``: ``HIPSolver``.

Same constructor, same ``setup(...) -> self``, ``reset_initial_seed``, ``reset_parameters``, ``solve``,
``stats``, ``did_solve``, ``number_of_iterations`` and the same ``error_on_fail`` -> ``RuntimeError``
behaviour (solver.py:133-134).  Arrays are numpy instead of ``casadi.DM``.  Additive extension: a
leading batch axis (``reset_parameters_batch`` / ``reset_initial_seed_batch`` / ``solve_batch``) for B
independent instances solved in one launch sequence; B=1 through the batch calls equals the scalar
interface.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from .backend import BatchResult, FigureEightBackend, IKBackend, MultiArmBackend, PointMassBackend, QPBackend, TapeBackend, TorqueBackend, tape_backend, tape_default_max_iter
from .lowering import FigureEightSpec, IkSpec, MultiArmSpec, PointMassSpec, QpSpec, TapeSpec, TorqueSpec, lower
from .models import RobotModel
from .optimization import Optimization


class Solver(ABC):
    """solver.py:61-314 (numpy in place of casadi.DM)."""

    def __init__(self, optimization: Optimization, error_on_fail: bool = False):
        self.opt = optimization
        self.x0 = np.zeros(optimization.nx)  # solver.py:76
        self.p = np.zeros(optimization.np)  # solver.py:79
        self._p_dict: Dict[str, np.ndarray] = {}
        self._error_on_fail = error_on_fail
        self._solution = None

    @property
    def opt_type(self) -> type:
        return type(self.opt)

    @abstractmethod
    def setup(self, *args, **kwargs):
        pass

    def reset_initial_seed(self, x0: Dict[str, np.ndarray]) -> None:
        self.x0 = self.opt.decision_variables.dict2vec(x0)

    def reset_parameters(self, p: Dict[str, np.ndarray]) -> None:
        self.p = self.opt.parameters.dict2vec(p)
        self._p_dict = self.opt.parameters.vec2dict(self.p)

    @abstractmethod
    def _solve(self) -> np.ndarray:
        pass

    def _add_model_states(self, solution: dict, p_dict: dict) -> dict:
        # solver.py:136-155
        for model in self.opt.models:
            for d in model.time_derivs:
                n_s = model.state_name(d)
                n_s_x = model.state_optimized_name(d)
                if isinstance(model, RobotModel) and model.num_param_joints > 0:
                    n_s_p = model.state_parameter_name(d)
                    t = solution[n_s_x].shape[1]
                    full = np.zeros((model.dim, t))
                    full[model.optimized_joint_indexes, :] = solution[n_s_x]
                    full[model.parameter_joint_indexes, :] = p_dict[n_s_p]
                    solution[n_s] = full
                else:
                    solution[n_s] = solution[n_s_x]
        return solution

    def solve(self) -> Dict[str, np.ndarray]:
        solution = self.opt.decision_variables.vec2dict(self._solve())
        if self._error_on_fail and (not self.did_solve()):
            raise RuntimeError("Solver failed!")
        return self._add_model_states(solution, self._p_dict)

    @abstractmethod
    def stats(self):
        pass

    @abstractmethod
    def did_solve(self) -> bool:
        pass

    @abstractmethod
    def number_of_iterations(self) -> int:
        pass

    # ---- diagnostics (solver.py:167-237, 269-314), evaluated by optas_amd.evaluate (FK through liboptas_hip) ----
    def evaluate_cost_terms(self, x: Dict[str, np.ndarray], p: Dict[str, np.ndarray]) -> List:
        """Value of each cost term at (x, p), in cost_terms order (solver.py:282-314)."""
        from .evaluate import evaluate

        xv = self.opt.decision_variables.dict2vec(x)
        pv = self.opt.parameters.dict2vec(p)
        return [float(evaluate(term, self.opt, xv, pv).reshape(-1)[0]) for term in self.opt.cost_terms.values()]

    def evaluate_cost(self, x: Dict[str, np.ndarray], p: Dict[str, np.ndarray]) -> float:
        """f(x, p) = sum of the cost terms (solver.py:269-280; optimization.py:192-195)."""
        return float(sum(self.evaluate_cost_terms(x, p)))

    def violated_constraints(self, x: Dict[str, np.ndarray], p: Dict[str, np.ndarray]):
        """Per constraint block: (label, ctype, diff, pattern) with diff = rhs - lhs evaluated at (x, p) and
        pattern = diff >= 0, in the reference's order lin_eq, eq, lin_ineq, ineq (solver.py:167-237 -- including its
        quirk that the pattern marks the *satisfied* rows)."""
        from dataclasses import dataclass

        from .evaluate import evaluate

        @dataclass
        class ViolatedConstraint:
            label: str
            ctype: str
            diff: np.ndarray
            pattern: np.ndarray

            def __str__(self):
                return f"\n{self.label} [{self.ctype}]:\n{self.pattern}\n"

        xv = self.opt.decision_variables.dict2vec(x)
        pv = self.opt.parameters.dict2vec(p)
        out = []
        for ctype, cont in (("lin_eq", self.opt.lin_eq_constraints), ("eq", self.opt.eq_constraints),
                            ("lin_ineq", self.opt.lin_ineq_constraints), ("ineq", self.opt.ineq_constraints)):
            lst = []
            for label, term in cont.items():
                diff = evaluate(term, self.opt, xv, pv)
                lst.append(ViolatedConstraint(label, ctype, diff, diff >= 0.0))
            out.append(lst)
        return tuple(out)

    @staticmethod
    def interpolate(traj, T: float, **interp_args):
        """solver.py:239-251."""
        from scipy.interpolate import interp1d

        traj = np.asarray(traj, dtype=np.float64)
        t = np.linspace(0, T, traj.shape[1])
        return interp1d(t, traj, **interp_args)


class _LeadAdapter:
    """Figure-eight family with one parameterised joint ahead of the chain (RobotModel(param_joints=[...]),
    example/figure_eight_plan_6dof.py): re-packs the reference's parameter vector ["{name}/q/p" (1 x T); "{name}/dq/p" (1 x (T-1)); qc (ndof)]
    into the kernel's row [qc of the optimised joints; lead angle of qc; lead angle per knot] and adds the constant
    w_vel * ||dq/p||^2 that the reference's joint-velocity cost carries for the parameterised row."""

    def __init__(self, opt, spec, backend):
        self.opt, self.spec, self.be = opt, spec, backend

    def solve(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        p = np.asarray(p, dtype=np.float64).reshape(-1, int(self.opt.np))
        off, o_ = {}, 0
        for k_, v_ in self.opt.parameters.items():  # (label -> offset from the items' shapes: the reference's SXContainer has no offsets())
            shp = tuple(getattr(v_, "shape", None) or v_.size())
            off[k_] = o_
            o_ += int(shp[0]) * int(shp[1])
        T, sp = self.spec.T, self.spec
        qc = p[:, off[sp.qc_name] : off[sp.qc_name] + sp.robot.ndof]
        qp = p[:, off[sp.lead["qp"]] : off[sp.lead["qp"]] + T]
        dqp = p[:, off[sp.lead["dqp"]] : off[sp.lead["dqp"]] + T - 1]
        # the kernels eliminate knots 0 and 1 at the fixed configuration qc (q_0 = qc, dq_0 = 0): their lead angles are taken to be qc's
        lead_c = qc[:, [sp.lead["par"]]]
        if np.abs(qp[:, :2] - lead_c).max() > 1e-12:
            raise ValueError(f"'{sp.lead['qp']}': the parameterised joint must sit at its value in '{sp.qc_name}' on the first two knots "
                             "(the fixed knots of the problem)")
        pk = np.ascontiguousarray(np.concatenate([qc[:, sp.lead["opt"]], lead_c, qp], axis=1))
        r = self.be.solve(x0, pk)
        r.f = r.f + sp.w_vel * np.sum(dqp * dqp, axis=1)
        return r

    def multipliers(self, B: int):
        return self.be.multipliers(B)

    def timing(self) -> dict:
        return self.be.timing()

    def close(self) -> None:
        self.be.close()


class _PlannerAdapter:
    """example/point_mass_planner.py on the point-mass kernel family: p = [init; goal] becomes the family's row
    [curr = init; dcurr = 0; goal repeated on every knot (only the last one carries weight); the constant obstacle on every knot]."""

    def __init__(self, spec, backend):
        self.spec, self.be = spec, backend

    def solve(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        p = np.asarray(p, dtype=np.float64).reshape(-1, 4)
        B, T = p.shape[0], self.spec.T
        rows = np.concatenate([p[:, :2], np.zeros((B, 2)), np.tile(p[:, 2:4], (1, T)), np.tile(self.spec.planner["obstacle"], (B, T))], axis=1)
        return self.be.solve(x0, np.ascontiguousarray(rows))

    def solve_ms(self) -> float:
        return self.be.solve_ms()

    def close(self) -> None:
        self.be.close()


class _QpAdapter:
    """Dense QP family: reads P, q, M, c, A, b off the Optimization's numeric members for every instance (they may all depend on the
    parameters) and hands [P | q | M | c | A | b] rows to the kernel.  The cost's constant term f(0, p) is added back to f."""

    def __init__(self, opt, backend: QPBackend):
        self.opt, self.be = opt, backend

    def solve(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1, self.opt.nx)
        p = np.asarray(p, dtype=np.float64).reshape(x0.shape[0], -1)
        o = self.opt
        if self.be.tape is not None:  # the device reads the QP off the problem's tape (oh_qp_set_tape): no per-instance host work
            return self.be.solve(x0, p)
        z = np.zeros(o.nx)
        rows, f0 = [], []
        for pb in p:
            M = o.M(pb) if o.nk else np.zeros((0, o.nx))
            c = o.c(pb) if o.nk else np.zeros(0)
            A = o.A(pb) if o.na else np.zeros((0, o.nx))
            b = o.b(pb) if o.na else np.zeros(0)
            rows.append(QPBackend.pack(o.P(pb), o.q(pb), M, c, A, b))
            f0.append(o.f(z, pb))
        r = self.be.solve(x0, np.stack(rows))
        r.f = r.f + np.asarray(f0)
        return r

    def close(self) -> None:
        self.be.close()


def figure_eight_backend(spec: FigureEightSpec, o: dict, hessian: int) -> FigureEightBackend:
    """The OH_PROBLEM_FIGURE_EIGHT handle of a lowered problem (shared by HIPSolver.setup and the literal optas.solver.Solver subclass of
    optas_amd.casadi_tape); consumes its options from ``o``."""
    chain = spec.robot.solver_chain(spec.link)
    guards = None
    if spec.lo is not None or spec.spheres is not None or spec.vlo is not None:
        guards = _lib.oh_guards()
    if spec.vlo is not None:
        guards.vel_limits = 1
        for j in range(spec.robot.ndof):
            guards.dq_lo[j], guards.dq_up[j] = float(spec.vlo[j]), float(spec.vup[j])
    if spec.lo is not None:
        guards.limits = 1
        for j in range(spec.robot.ndof):
            guards.q_lo[j], guards.q_up[j] = float(spec.lo[j]), float(spec.up[j])
    if spec.spheres is not None:
        guards.n_links, guards.n_obstacles = len(spec.spheres.links), len(spec.spheres.obstacles)
        for l, (k, off) in enumerate(spec.robot.link_attachments(spec.link, spec.spheres.links)):
            if k < 0:
                raise NotImplementedError(f"sphere link '{spec.spheres.links[l]}' does not move with any joint of the chain")
            guards.link_joint[l] = k
            for i in range(3):
                guards.link_offset[l][i] = float(off[i])
    return FigureEightBackend(
        chain,
        spec.T,
        spec.dt,
        spec.local_path,
        w_path=spec.w_path,
        w_vel=spec.w_vel,
        max_iter=int(o.pop("max_iter", 200)),
        tol=float(o.pop("tol", 1e-6)),
        tol_feas=float(o.pop("tol_feas", 1e-9)),
        hessian=hessian,
        mu0=float(o.pop("mu0", 0.0)),
        guards=guards,
    )


class HIPSolver(Solver):
    """MI355X backend.  ``setup(solver_options)`` lowers the problem to a kernel family
    (optas_amd.lowering) and creates the liboptas_hip handle; it raises if the problem is not lowerable
    or if no HIP device is present -- there is no CPU path."""

    def setup(self, solver_name: str = "hip_sqp", solver_options: Optional[Dict] = None):
        if solver_name != "hip_sqp":
            raise ValueError(f"solver '{solver_name}' does not support this problem type")  # solver.py:371-373
        o = dict(solver_options or {})
        # options of the library handle(s) behind this solver (include/optas_hip.h: oh_set_option), e.g. {"batch_invariant": 1}
        handle_options = dict(o.pop("options", None) or {})
        kind, spec = lower(self.opt)
        self._kind, self._spec = kind, spec
        hessian = {"gauss_newton": _lib.OH_HESSIAN_GAUSS_NEWTON, "exact": _lib.OH_HESSIAN_EXACT, "hybrid": _lib.OH_HESSIAN_HYBRID}[o.get("hessian", "hybrid")]
        if isinstance(spec, FigureEightSpec):
            o.pop("hessian", None)
        if isinstance(spec, FigureEightSpec):
            self._backend = figure_eight_backend(spec, o, hessian)
            if spec.lead is not None:
                self._backend = _LeadAdapter(self.opt, spec, self._backend)
        elif isinstance(spec, TorqueSpec):
            o.pop("hessian", None)
            # (round 3 names of the augmented-Lagrangian solver this family had before the interior point: accepted, ignored, said so -- ADVICE r4)
            for old, new in (("tol_feas", "tol_compl"), ("rho0", "mu_barrier0")):
                if old in o:
                    import warnings

                    warnings.warn(f"hip_sqp / torque MPC: option '{old}' belonged to the round-3 augmented-Lagrangian solver and is ignored; the interior point "
                                  f"takes '{new}'", DeprecationWarning, stacklevel=2)
                    o.pop(old)
            self._backend = TorqueBackend(spec.robot.solver_chain(spec.link), spec.robot.dynamics_tables(), T=spec.T, dt=spec.dt, w_path=spec.w_path,
                                          w_vel=spec.w_vel, w_tau=spec.w_tau, tau_lo=spec.tau_lo, tau_up=spec.tau_up, dq_lo=getattr(spec, "dq_lo", None), dq_up=getattr(spec, "dq_up", None),
                                          max_iter=int(o.pop("max_iter", 300)), tol=float(o.pop("tol", 1e-6)), tol_compl=float(o.pop("tol_compl", 1e-8)),
                                          mu_barrier0=float(o.pop("mu_barrier0", 0.0)), mu0=float(o.pop("mu0", 0.0)))
        elif isinstance(spec, PointMassSpec):
            o.pop("hessian", None)
            pl = spec.planner
            self._backend = PointMassBackend(
                spec.T, spec.dt, spec.w_acc, spec.ylim, spec.vlim, spec.safe, max_iter=int(o.pop("max_iter", 200 if pl else 100)),
                tol=float(o.pop("tol", 1e-8)), track_final_only=pl is not None, w_vel=pl["w_vel"] if pl else 0.0, fix_final_velocity=pl is not None,
            )
            if pl is not None:
                self._backend = _PlannerAdapter(spec, self._backend)
        elif isinstance(spec, MultiArmSpec):
            o.pop("hessian", None)
            self._backend = MultiArmBackend(spec, self.opt, max_iter=int(o.pop("max_iter", 200)), tol=float(o.pop("tol", 1e-6)), hessian=hessian)
        elif isinstance(spec, IkSpec):
            o.pop("hessian", None)
            self._backend = IKBackend(spec.robot.kinematic_chain(spec.link), spec.lo, spec.up, w_nominal=spec.w_nominal,
                                      max_iter=int(o.pop("max_iter", 200)), tol=float(o.pop("tol", 1e-6)), tol_feas=float(o.pop("tol_feas", 1e-9)))
        elif isinstance(spec, QpSpec):
            o.pop("hessian", None)
            qb = QPBackend(spec.n, spec.m, spec.me, max_iter=int(o.pop("max_iter", 100)), tol=float(o.pop("tol", 1e-9)))
            if bool(o.pop("device_assembly", True)):
                # P, q, M, c, A, b on the device from the problem's instruction tape; problems the tape compiler does not take (too long,
                # an expression node without a scalar expansion) keep the host route, where the numeric members are probed per instance
                try:
                    from .tape import compile_problem

                    qb.set_tape(compile_problem(spec.problem or self.opt))
                except (NotImplementedError, TypeError, ValueError):
                    pass
            self._backend = _QpAdapter(spec.problem or self.opt, qb)  # spec.problem: band rows rewritten as linear pairs (lowering._band_rows)
        elif isinstance(spec, TapeSpec):
            o.pop("hessian", None)
            # (evaluations: a small dense problem needs a few hundred; the limited-memory path of a trajectory-sized one tens of thousands)
            # ("eliminate": affine equality rows -- Euler rows, pinned configurations -- are substituted away before the tape reaches the GPU, tape.py)
            # (handle options that shape the evaluator -- tape_wave, tape_lbfgs, ... -- go in at creation: applied afterwards they rebuilt the evaluator on a tape
            #  already rebalanced for the wavefront path, left backend.wave / backend.jit stale and overrode tape_backend's own tape_lbfgs default)
            tape_opts = {k: handle_options.pop(k) for k in [k for k in handle_options if k.startswith("tape_")]}
            self._backend = tape_backend(spec.tape, options=tape_opts or None, eliminate=bool(o.pop("eliminate", True)), max_iter=int(o.pop("max_iter", tape_default_max_iter(spec.tape.nx))),
                                         tol=float(o.pop("tol", 1e-6)), tol_feas=float(o.pop("tol_feas", 1e-9)), rho0=(float(o.pop("rho0")) if "rho0" in o else None), jit=bool(o.pop("jit", True)), metric=o.pop("metric", None))
        else:  # pragma: no cover
            raise NotImplementedError(kind)
        if o:
            raise ValueError(f"unknown solver options {sorted(o)}")
        if handle_options:
            target = self._backend
            while not hasattr(target, "set_options") and hasattr(target, "be"):  # the adapters keep their backend in .be
                target = target.be
            target.set_options(handle_options)
        self._stats: Optional[dict] = None
        self._x0_batch: Optional[np.ndarray] = None
        self._p_batch: Optional[np.ndarray] = None
        return self

    # ---- scalar interface (one instance), solver.py:103-157 ---------------------------------------------
    def _solve(self) -> np.ndarray:
        res = self._backend.solve(self.x0.reshape(1, -1), self.p.reshape(1, -1))
        self._record(res)
        return res.x[0]

    def _record(self, res: BatchResult) -> None:
        self._solution = res
        self._stats = {
            "success": bool(np.all(_lib.status_ok(res.status))),
            "iter_count": int(res.iters.max()),
            "return_status": [_lib.STATUS_NAMES.get(int(s), "Unknown") for s in res.status],
            "f": res.f.copy(),
            "kkt": res.kkt.copy(),
            "iterations": res.iters.copy(),
            "status": res.status.copy(),
            "solution": res,
        }

    def stats(self) -> dict:
        return self._stats

    def did_solve(self) -> bool:
        return self._stats["success"]  # solver.py:407-412

    def number_of_iterations(self) -> int:
        return self._stats["iter_count"]  # solver.py:414-419

    # ---- batch extension ------------------------------------------------------------------------------------
    def reset_initial_seed_batch(self, x0: Dict[str, np.ndarray]) -> None:
        """Each value has a leading batch axis: (B, m, n)."""
        B = len(next(iter(x0.values()))) if x0 else 1
        self._x0_batch = self.opt.decision_variables.dict2vec_batch(x0, B)

    def reset_parameters_batch(self, p: Dict[str, np.ndarray]) -> None:
        B = len(next(iter(p.values())))
        self._p_batch = self.opt.parameters.dict2vec_batch(p, B)

    def solve_batch(self, stacked: bool = False):
        """B instances in one launch sequence.  Returns a list of B solution dicts shaped like ``solve()``'s, or, with
        ``stacked=True``, one dict of arrays with a leading batch axis (no per-instance Python objects: the fast form for large B)."""
        assert self._p_batch is not None, "call reset_parameters_batch first"
        B = self._p_batch.shape[0]
        x0 = self._x0_batch if self._x0_batch is not None else np.zeros((B, self.opt.nx))
        assert x0.shape[0] == B, "seed and parameter batches differ in size"
        res = self._backend.solve(x0, self._p_batch)
        self._record(res)
        if self._error_on_fail and (not self.did_solve()):
            raise RuntimeError("Solver failed!")
        sol = self.opt.decision_variables.vec2dict_batch(res.x)
        pd = self.opt.parameters.vec2dict_batch(self._p_batch)
        # "{name}/{d}q" full states (solver.py:136-155), batched
        for model in self.opt.models or []:
            for d in model.time_derivs:
                n_s, n_s_x = model.state_name(d), model.state_optimized_name(d)
                if isinstance(model, RobotModel) and model.num_param_joints > 0:
                    t = sol[n_s_x].shape[2]
                    full = np.zeros((B, model.dim, t))
                    full[:, model.optimized_joint_indexes, :] = sol[n_s_x]
                    full[:, model.parameter_joint_indexes, :] = pd[model.state_parameter_name(d)]
                    sol[n_s] = full
                else:
                    sol[n_s] = sol[n_s_x]
        if stacked:
            return sol
        return [{k: v[b] for k, v in sol.items()} for b in range(B)]

    def solve_batch_arrays(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        """Array-in/array-out fast path (no dict shuffling): x0 (B, nx), p (B, np) in vec() order."""
        res = self._backend.solve(x0, p)
        self._record(res)
        return res

    @property
    def backend(self):
        return self._backend
