"""Thin object wrapper over the liboptas_hip handle for one lowered problem (ctypes, numpy buffers).

``HIPSolver`` (optas_amd/solver.py) is the user-facing, reference-shaped class; this is the piece
that owns the ``oh_handle`` and moves arrays across the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib


@dataclass
class BatchResult:
    x: np.ndarray  # (B, nx) reference layout
    f: np.ndarray  # (B,)
    kkt: np.ndarray  # (B, 3) stationarity, feasibility, complementarity
    iters: np.ndarray  # (B,)
    status: np.ndarray  # (B,)


class _OptionsMixin:
    """Per-handle options by name (oh_set_option; include/optas_hip.h lists them)."""

    def set_option(self, name: str, value: float):
        _lib.set_option(self._h, name, value)
        return self

    def set_options(self, options=None, **kw):
        for k, v in {**(options or {}), **kw}.items():
            self.set_option(k, v)
        return self

    def get_option(self, name: str) -> float:
        return _lib.get_option(self._h, name)


class _SolveMixin(_OptionsMixin):
    """oh_solve / oh_solve_device / timing over an existing handle (self._h, self.nx, self.np_)."""

    def solve(self, x0: np.ndarray, p: np.ndarray) -> "BatchResult":
        lib = _lib.load()
        x0 = _lib.as_f64(x0)
        p = _lib.as_f64(p)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        B = x0.shape[0]
        assert x0.shape == (B, self.nx), f"x0 must be (B, {self.nx})"
        assert p.shape == (B, self.np_), f"p must be (B, {self.np_})"
        chunk = getattr(self, "max_batch", None)  # one oh_solve call of the locked trajectory family is bounded (optas_hip.h)
        if chunk and B > chunk:
            parts = [self.solve(x0[i : i + chunk], p[i : i + chunk]) for i in range(0, B, chunk)]
            return BatchResult(*(np.concatenate([getattr(r, k) for r in parts]) for k in ("x", "f", "kkt", "iters", "status")))
        x = np.empty((B, self.nx))
        f = np.empty(B)
        kkt = np.empty((B, 3))
        iters = np.empty(B, dtype=np.int32)
        status = np.empty(B, dtype=np.int32)
        _lib.check(
            lib.oh_solve(self._h, B, _lib._ptr(x0), _lib._ptr(p), _lib._ptr(x), _lib._ptr(f), _lib._ptr(kkt), _lib._ptr(iters), _lib._ptr(status)),
            "oh_solve",
        )
        return BatchResult(x, f, kkt, iters, status)

    def solve_device(self, B: int, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status) -> None:
        g = lambda b: None if b is None else b.ptr
        _lib.check(
            _lib.load().oh_solve_device(self._h, int(B), g(d_x0), g(d_p), g(d_x), g(d_f), g(d_kkt), g(d_iters), g(d_status)),
            "oh_solve_device",
        )

    def solve_ms(self) -> float:
        out = (C.c_double * 11)()
        _lib.check(_lib.load().oh_get_timing(self._h, out), "oh_get_timing")
        return out[4]

    def set_profiling(self, on: bool) -> None:
        """oh_set_profiling: families that run several launches per solve record one HIP event after every kernel (one stream, no split)."""
        _lib.check(_lib.load().oh_set_profiling(self._h, 1 if on else 0), "oh_set_profiling")

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().oh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PointMassBackend(_SolveMixin):
    """OH_PROBLEM_POINT_MASS_MPC handle (example/point_mass_mpc.py Controller)."""

    def __init__(self, T=20, dt=0.05, w_acc=0.0025 / 20, ylim=1.5, vlim=1.0, safe=0.3, max_iter=100, tol=1e-8, track_final_only=False, w_vel=0.0,
                 fix_final_velocity=False):
        lib = _lib.load()
        self.T = int(T)
        self.nx, self.np_ = 4 * self.T, 4 + 4 * self.T
        desc = _lib.oh_pointmass_desc(T=self.T, dt=float(dt), w_acc=float(w_acc), ylim=float(ylim), vlim=float(vlim), safe=float(safe),
                                      max_iter=int(max_iter), tol=float(tol), track_final_only=1 if track_final_only else 0, w_vel=float(w_vel),
                                      fix_final_velocity=1 if fix_final_velocity else 0)
        self._h = C.c_void_p()
        _lib.check(lib.oh_create_pointmass(C.byref(desc), C.byref(self._h)), "oh_create_pointmass")

    def rollout(self, state0: np.ndarray, obs_table: np.ndarray, n_ticks: int, advance: int = 2, ramp: float = 0.032):
        """Closed-loop receding horizon on the device (oh_pm_rollout): state0 (B, 4) = (y, dy); obs_table (n_ticks*advance + T, 2).
        Returns states (n_ticks + 1, B, 4), f, iters, status (n_ticks, B)."""
        state0 = _lib.as_f64(state0).reshape(-1, 4)
        B = state0.shape[0]
        need = n_ticks * advance + self.T
        obs_table = _lib.as_f64(obs_table).reshape(-1, 2)
        assert obs_table.shape[0] >= need, f"obs_table needs {need} rows"
        obs_table = np.ascontiguousarray(obs_table[:need])
        states = np.empty((n_ticks + 1, B, 4))
        f = np.empty((n_ticks, B))
        iters = np.empty((n_ticks, B), dtype=np.int32)
        status = np.empty((n_ticks, B), dtype=np.int32)
        _lib.check(
            _lib.load().oh_pm_rollout(self._h, B, int(n_ticks), int(advance), float(ramp), _lib._ptr(state0), _lib._ptr(obs_table), _lib._ptr(states),
                                      _lib._ptr(f), _lib._ptr(iters), _lib._ptr(status)),
            "oh_pm_rollout",
        )
        return states, f, iters, status


class TapeBackend(_SolveMixin):
    """OH_PROBLEM_TAPE handle: a compiled instruction tape (optas_amd.tape.Tape) interpreted on the GPU; x (B, nx), p (B, np)."""

    def __init__(self, tape, max_iter=2000, tol=1e-6, tol_feas=1e-9, rho0=10.0, jit=True, wave=True, options=None, keep_regs=None, metric=False):
        """metric (default False since round 6: on a problem as written the penalty on hundreds of affine rows is most of the merit's curvature and none of it is
        in the cost's block -- tape_backend(), which knows whether the affine rows are gone, makes the choice): in the limited-memory regime (beyond 48 variables) hand the library the inverse of the constant block of the cost's Hessian as the
        initial metric of the quasi-Newton iteration (tape.py:quadratic_cost_metric, oh_tape_set_metric) where the cost has one; wave: let trajectory-sized tapes (beyond 48 variables) run one block of wavefronts per instance over the dependency levels of the tape
        (csrc/oh_tape_wave.hip) where the library finds that it applies; options: oh_set_option pairs applied to the handle; keep_regs: registers the
        caller will read back with probe() -- self.kept_regs holds their indices in the tape the handle was given (re-association renumbers)."""
        self.nx, self.np_ = int(tape.nx), max(1, int(tape.np_))
        self.kept_regs = None if keep_regs is None else np.asarray(keep_regs, dtype=np.int32)
        self._np_real = int(tape.np_)
        self._h = None
        self._h0 = None
        if isinstance(metric, np.ndarray):  # (computed by the caller: tape_backend chooses the initial penalty by it)
            self._h0 = np.ascontiguousarray(metric, dtype=np.float64).reshape(self.nx, self.nx)
        elif metric and int(tape.nx) > 48 and int((options or {}).get("tape_lbfgs", -1)) != 0:  # (tape_lbfgs 0 forces the dense form, which builds its own matrix)
            from .tape import quadratic_cost_metric

            self._h0 = quadratic_cost_metric(tape)
        want_wave = bool(wave) and int(tape.nx) > 48 and float((options or {}).get("tape_wave", 1)) != 0.0
        if want_wave:
            # chains of additions are dependency levels for that evaluator, so sums go in as balanced trees (same values to the rounding of the
            # summation order).  Whether the path is taken is the library's decision (limited-memory regime, LDS fit): the handle is asked, and a
            # tape the library declined is handed over again as it was written (one gate, not an environment variable read twice)
            from .tape import Tape, rebalance_sums

            if keep_regs is None:
                bal, kept = rebalance_sums(tape), None
            else:  # the registers to keep ride along as extra outputs of the re-association, then leave the row list again
                nk = len(self.kept_regs)
                tmp = rebalance_sums(Tape(tape.op, tape.a, tape.b, tape.c, tape.out_cost, np.concatenate([tape.out_rows, self.kept_regs]).astype(np.int32),
                                          tape.n_ineq, tape.n_eq + nk, tape.nx, tape.np_))
                bal = Tape(tmp.op, tmp.a, tmp.b, tmp.c, tmp.out_cost, tmp.out_rows[: len(tmp.out_rows) - nk].copy(), tape.n_ineq, tape.n_eq, tape.nx, tape.np_)
                kept = np.asarray(tmp.out_rows[len(tmp.out_rows) - nk :], dtype=np.int32)
            self._create(bal, max_iter, tol, tol_feas, rho0, jit, options)
            if self.flag("tape_wave") == 0:
                self.close()
                want_wave = False
            elif kept is not None:
                self.kept_regs = kept
        if not want_wave:
            opts = dict(options or {})
            if int(tape.nx) > 48:
                opts["tape_wave"] = 0
            self._create(tape, max_iter, tol, tol_feas, rho0, jit, opts)
        self.wave = self.flag("tape_wave") != 0
        self.jit = bool(jit) and not self.wave  # what actually runs: generated code only where the wavefront evaluator does not

    def _create(self, tape, max_iter, tol, tol_feas, rho0, jit, options):
        lib = _lib.load()
        self.tape = tape
        self._keep = [np.ascontiguousarray(tape.op, dtype=np.int32), np.ascontiguousarray(tape.a, dtype=np.int32), np.ascontiguousarray(tape.b, dtype=np.int32),
                      np.ascontiguousarray(tape.c, dtype=np.float64), np.ascontiguousarray(np.append(tape.out_rows, 0), dtype=np.int32)]
        opts = dict(options or {})
        # the two choices that shape the evaluator travel in the descriptor (it is built when the handle is created)
        desc = self.descriptor(tape, self._keep, max_iter, tol, tol_feas, rho0, jit, no_wave=float(opts.pop("tape_wave", 1)) == 0.0, lbfgs=int(opts.pop("tape_lbfgs", -1)))
        self._h = C.c_void_p()
        _lib.check(lib.oh_create_tape(C.byref(desc), C.byref(self._h)), "oh_create_tape")
        if self._h0 is not None:
            _lib.check(lib.oh_tape_set_metric(self._h, _lib._ptr(self._h0)), "oh_tape_set_metric")
        if opts:
            self.set_options(opts)

    @staticmethod
    def descriptor(tape, keep=None, max_iter=2000, tol=1e-6, tol_feas=1e-9, rho0=10.0, jit=True, no_wave=False, lbfgs=-1):
        """oh_tape_desc over the arrays of a compiled tape; `keep` receives the contiguous arrays the descriptor points into."""
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        if not keep:
            keep = keep if keep is not None else []
            keep += [np.ascontiguousarray(tape.op, dtype=np.int32), np.ascontiguousarray(tape.a, dtype=np.int32), np.ascontiguousarray(tape.b, dtype=np.int32),
                     np.ascontiguousarray(tape.c, dtype=np.float64), np.ascontiguousarray(np.append(tape.out_rows, 0), dtype=np.int32)]
        desc = _lib.oh_tape_desc(nx=int(tape.nx), np=int(tape.np_), len=len(tape.op), op=keep[0].ctypes.data_as(ip), a=keep[1].ctypes.data_as(ip),
                                 b=keep[2].ctypes.data_as(ip), c=keep[3].ctypes.data_as(dp), out_cost=int(tape.out_cost), n_ineq=int(tape.n_ineq),
                                 n_eq=int(tape.n_eq), rows=keep[4].ctypes.data_as(ip), max_iter=int(max_iter), tol=float(tol), tol_feas=float(tol_feas),
                                 rho0=float(rho0), jit=int(bool(jit)), no_wave=int(bool(no_wave)), lbfgs=(0 if lbfgs < 0 else (int(lbfgs) if lbfgs > 0 else -1)))
        desc._keep = keep
        return desc

    @staticmethod
    def generated_source(tape):
        """(source text, code-object bytes) of the kernel hiprtc builds for this tape; needs no GPU (oh_tape_compile)."""
        desc = TapeBackend.descriptor(tape)
        size, n = C.c_size_t(0), C.c_size_t(0)
        _lib.check(_lib.load().oh_tape_compile(C.byref(desc), C.byref(size), None, 0, C.byref(n)), "oh_tape_compile")
        buf = C.create_string_buffer(n.value + 1)
        _lib.check(_lib.load().oh_tape_compile(C.byref(desc), C.byref(size), buf, n.value + 1, C.byref(n)), "oh_tape_compile")
        return buf.value.decode(), int(size.value)

    def probe(self, x, p, regs=None, seeds=None):
        """oh_tape_probe: values of `regs` at the points x (B, nx), p (B, np); with seeds (B, 1 + n_ineq + n_eq) = weights of (cost, rows) also the
        derivative of that combination with respect to each of the registers and its gradient with respect to x.  Returns (val, adj, grad)."""
        x = _lib.as_f64(x).reshape(-1, self.nx)
        B = x.shape[0]
        p = _lib.as_f64(p).reshape(B, -1) if self._np_real else np.zeros((B, 1))
        regs = np.ascontiguousarray(np.zeros(0) if regs is None else regs, dtype=np.int32)
        val = np.empty((B, len(regs)))
        adj = grad = None
        if seeds is not None:
            seeds = _lib.as_f64(seeds).reshape(B, 1 + int(self.tape.n_ineq) + int(self.tape.n_eq))
            adj, grad = np.empty((B, len(regs))), np.empty((B, self.nx))
        _lib.check(_lib.load().oh_tape_probe(self._h, B, _lib._ptr(x), _lib._ptr(p), len(regs), _lib._ptr(regs) if len(regs) else None, _lib._ptr(val) if len(regs) else None,
                                             _lib._ptr(seeds), _lib._ptr(adj) if (adj is not None and len(regs)) else None, _lib._ptr(grad)), "oh_tape_probe")
        return val, adj, grad

    def flag(self, name: str) -> int:
        """oh_get_flag: 'tape_wave' (0 thread per instance, 1 / 2 wavefront per instance), 'tape_regs_lds', 'tape_levels', 'tape_passes', 'tape_metric'."""
        v = C.c_int(0)
        _lib.check(_lib.load().oh_get_flag(self._h, name.encode(), C.byref(v)), "oh_get_flag")
        return int(v.value)

    def solve(self, x0, p):
        p = _lib.as_f64(p).reshape(len(np.atleast_2d(x0)), -1)
        if self._np_real == 0:
            p = np.zeros((p.shape[0], 1))
        return super().solve(x0, p)

    def multipliers(self, B: int):
        ni, ne = int(self.tape.n_ineq), int(self.tape.n_eq)
        out = np.empty((B, ni + ne))
        if ni + ne:
            _lib.check(_lib.load().oh_get_multipliers(self._h, int(B), _lib._ptr(out)), "oh_get_multipliers")
        return out[:, :ni], out[:, ni:]


class EliminatedTapeBackend:
    """A generic problem whose affine equality rows -- the Euler rows of integrate_model_states, fix_configuration, initial_configuration
    (builder.py:419-469, 511-539) -- have been eliminated by substitution (optas_amd/tape.py:eliminate_affine_equalities): the GPU solves over the free
    variables, and the eliminated variables and the multipliers of their rows are read back from the device (oh_tape_probe: one forward and one reverse
    sweep at the solution).  Presents the ORIGINAL problem: x (B, nx), multipliers (B, n_ineq), (B, n_eq) in the original row order."""

    def __init__(self, tape, elim, **kw):
        self.full, self.elim = tape, elim
        self.nx, self.np_ = int(tape.nx), max(1, int(tape.np_))
        self.inner = TapeBackend(elim.tape, keep_regs=elim.def_regs, **kw)
        self.tape = tape
        self.jit, self.wave = self.inner.jit, self.inner.wave
        self._orig = None  # the problem as written, interpreter only: created when multipliers are asked for
        self._last = None

    def solve(self, x0, p):
        el = self.elim
        x0 = _lib.as_f64(x0).reshape(-1, self.nx)
        B = x0.shape[0]
        p = _lib.as_f64(p).reshape(B, -1)
        r = self.inner.solve(np.ascontiguousarray(x0[:, el.free]), p)
        val, _, _ = self.inner.probe(r.x, p, self.inner.kept_regs)  # the eliminated variables: registers of the reduced tape at the solution
        x = np.empty((B, self.nx))
        x[:, el.free] = r.x
        x[:, el.pivot] = val
        self._last = (x, p)
        return BatchResult(x, r.f, r.kkt, r.iters, r.status)

    def multipliers(self, B: int):
        """(lam (B, n_ineq), mu (B, n_eq)) in the ORIGINAL row order.  The rows that stayed carry the solver's multipliers; an eliminated row balances
        what is left of the stationarity condition in its pivot variable: with L' = f - lam^T g - mu_kept^T h, A_pivot^T nu = dL'/dx_pivot -- one
        reverse sweep of the tape as written at the solution (oh_tape_probe), then one constant triangular-sized solve per instance."""
        assert self._last is not None and len(self._last[0]) == B
        el, (x, p) = self.elim, self._last
        lam, mu = self.inner.multipliers(B)
        n_eq = int(self.full.n_eq)
        seeds = np.zeros((B, 1 + int(self.full.n_ineq) + n_eq))
        seeds[:, 0] = 1.0
        seeds[:, 1 : 1 + lam.shape[1]] = -lam
        seeds[:, 1 + lam.shape[1] + el.rows_kept] = -mu
        if self._orig is None:
            self._orig = TapeBackend(self.full, jit=False, wave=False, metric=False)
        _, _, grad = self._orig.probe(x, p, None, seeds)
        nu = np.linalg.solve(el.A_pivot.T, grad[:, el.pivot].T).T
        mu_full = np.zeros((B, n_eq))
        mu_full[:, el.rows_kept] = mu
        mu_full[:, el.rows_out] = nu
        return lam, mu_full

    def flag(self, name: str) -> int:
        return self.inner.flag(name)

    def timing(self) -> dict:
        return self.inner.timing()

    def solve_ms(self) -> float:
        return self.inner.solve_ms()

    def set_options(self, options=None, **kw):
        self.inner.set_options(options, **kw)
        return self

    def set_option(self, name, value):
        self.inner.set_option(name, value)
        return self

    def get_option(self, name):
        return self.inner.get_option(name)

    @property
    def handle(self):
        return self.inner.handle

    def close(self) -> None:
        self.inner.close()
        if self._orig is not None:
            self._orig.close()
            self._orig = None


def tape_backend(tape, eliminate=True, rho0=None, **kw):
    """TapeBackend for a compiled problem; trajectory-sized ones (beyond 48 variables: the limited-memory regime) first lose the equality rows that
    are affine in x with constant coefficients.  rho0 None: the initial penalty of the augmented Lagrangian is 10 for a problem as written, 1000 once its
    affine rows are gone, 1e4 if in addition its cost hands over a metric (below) (what is left are the few nonlinear rows; with 32 limited-memory pairs instead of 12 the planner takes 234-272 evaluations
    instead of 389-498, 256 instances 18.6 instead of 35.9 ms: tools/gpu_planner_sweep.py)."""
    metric = kw.pop("metric", None)  # None: where it pays -- on a problem whose affine rows are gone; True / an array: also on a problem as written; False: never
    if eliminate and int(tape.nx) > 48 and int(tape.n_eq) > 0:
        from .tape import eliminate_affine_equalities, quadratic_cost_metric

        el = eliminate_affine_equalities(tape)
        if el is not None and int(el.tape.nx) >= 1:
            opts = dict(kw.pop("options", None) or {})
            if int(el.tape.nx) > 48:
                opts.setdefault("tape_lbfgs", 32)
            # round 5, last part: where the cost of the reduced problem has a constant quadratic block (sumsqr terms on states, velocities, accelerations) its
            # inverse is the initial metric of the limited-memory iteration, and the penalty starts at 1e4: the pairs only have to learn the rows, and they
            # learn them once (planner, numpy port, 32 instances: 255 -> 54 evaluations on average, slowest 311 -> 84; tests/test_planner.py)
            h0 = None
            if metric is not False and int(el.tape.nx) > 48 and int(opts.get("tape_lbfgs", -1)) != 0:
                h0 = metric if isinstance(metric, np.ndarray) else quadratic_cost_metric(el.tape)
            if rho0 is None:
                rho0 = 1e4 if h0 is not None else 1000.0
            return EliminatedTapeBackend(tape, el, rho0=float(rho0), options=opts, metric=h0 if h0 is not None else False, **kw)
    # (as written, the penalty on hundreds of affine rows is most of the merit's curvature and none of it is in the cost's block: T = 60 planner, 840 variables,
    # 12 pairs -- one of two instances runs into the evaluation cap with the metric, none without; so only on request)
    return TapeBackend(tape, rho0=10.0 if rho0 is None else float(rho0), metric=False if metric is None else metric, **kw)


class QPBackend(_SolveMixin):
    """OH_PROBLEM_QP handle: x (B, n); p (B, n*n + n + m*n + m + me*n + me) = [P | q | M | c | A | b] per instance."""

    def __init__(self, n: int, m: int, me: int, max_iter=100, tol=1e-9, tape=None):
        """tape (optas_amd.tape.Tape of the problem, rows k then a): p of a solve is then (B, np of the problem) and the QP data is read off
        the tape on the device (oh_qp_set_tape) instead of arriving packed."""
        lib = _lib.load()
        self.n, self.m, self.me = int(n), int(m), int(me)
        self.nx = self.n
        self.np_ = self.n * self.n + self.n + self.m * self.n + self.m + self.me * self.n + self.me
        desc = _lib.oh_qp_desc(n=self.n, m=self.m, me=self.me, max_iter=int(max_iter), tol=float(tol))
        self._h = C.c_void_p()
        _lib.check(lib.oh_create_qp(C.byref(desc), C.byref(self._h)), "oh_create_qp")
        self.tape = None
        if tape is not None:
            self.set_tape(tape)

    def set_tape(self, tape) -> None:
        td = TapeBackend.descriptor(tape)
        _lib.check(_lib.load().oh_qp_set_tape(self._h, C.byref(td)), "oh_qp_set_tape")
        self.tape = tape
        self.np_ = max(1, int(tape.np_))

    def solve(self, x0, p):
        if self.tape is not None and int(self.tape.np_) == 0:  # a problem without parameters: the ABI still wants one column
            p = np.zeros((len(np.atleast_2d(x0)), 1))
        return super().solve(x0, p)

    @staticmethod
    def pack(P, q, M, c, A, b) -> np.ndarray:
        return np.concatenate([np.asarray(P, dtype=np.float64).reshape(-1), np.asarray(q, dtype=np.float64).reshape(-1),
                               np.asarray(M, dtype=np.float64).reshape(-1), np.asarray(c, dtype=np.float64).reshape(-1),
                               np.asarray(A, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)])

    def multipliers(self, B: int):
        out = np.empty((B, self.m + self.me))
        if self.m + self.me:
            _lib.check(_lib.load().oh_get_multipliers(self._h, int(B), _lib._ptr(out)), "oh_get_multipliers")
        return out[:, : self.m], out[:, self.m :]


class IKBackend(_SolveMixin):
    """OH_PROBLEM_IK handle (example/example.py): x = q, p = [q_nominal; p_goal]."""

    def __init__(self, chain: _lib.oh_chain, lo, up, w_nominal=1.0, max_iter=200, tol=1e-6, tol_feas=1e-9, rho0=0.0):
        lib = _lib.load()
        self.ndof = int(chain.ndof)
        self.nx, self.np_ = self.ndof, self.ndof + 3
        desc = _lib.oh_ik_desc(ndof=self.ndof, w_nominal=float(w_nominal), max_iter=int(max_iter), tol=float(tol), tol_feas=float(tol_feas),
                               rho0=float(rho0))
        lo, up = np.asarray(lo, dtype=np.float64).reshape(-1), np.asarray(up, dtype=np.float64).reshape(-1)
        assert lo.shape == (self.ndof,) and up.shape == (self.ndof,)
        for i in range(self.ndof):
            desc.q_lo[i], desc.q_up[i] = lo[i], up[i]
        self._h = C.c_void_p()
        _lib.check(lib.oh_create_ik(C.byref(desc), C.byref(self._h)), "oh_create_ik")
        _lib.check(lib.oh_set_constants(self._h, C.byref(chain)), "oh_set_constants")
        self.chain = chain

    def multipliers(self, B: int):
        """(mu_h (B,3), z_lo (B,ndof), z_up (B,ndof)) of the last solve, reference form."""
        out = np.empty((B, 3 + 2 * self.ndof))
        _lib.check(_lib.load().oh_get_multipliers(self._h, int(B), _lib._ptr(out)), "oh_get_multipliers")
        return out[:, :3], out[:, 3 : 3 + self.ndof], out[:, 3 + self.ndof :]


def tape_default_max_iter(nx: int) -> int:
    """Default evaluation budget of the generic tape family, shared by HIPSolver and the CasADi front end: a small dense problem needs a
    few hundred tape evaluations, the limited-memory path of a trajectory-sized one (nx > 48) tens of thousands."""
    return 2000 if nx <= 48 else 500000


class TorqueBackend(_SolveMixin):
    """OH_PROBLEM_TORQUE_MPC handle (BASELINE configs[4]): x = [vec(Q); vec(dQ); vec(ddQ); vec(TAU)], p = [qc; dqc; vec(goal 3 x T)]."""

    def __init__(self, chain: _lib.oh_chain, dynamics: _lib.oh_dynamics, T=30, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=None, tau_up=None,
                 max_iter=300, tol=1e-6, tol_compl=1e-8, mu_barrier0=0.0, mu0=0.0, dq_lo=None, dq_up=None):
        """dq_lo / dq_up: joint-velocity limits on the velocity states (enforce_model_limits(name, time_deriv=1)); None: no such rows.
        tol: reduced gradient of the Lagrangian; tol_compl: complementarity of the inequality rows (the barrier parameter the interior point ends at)."""
        lib = _lib.load()
        self.ndof, self.T = int(chain.ndof), int(T)
        self.nx, self.np_ = 4 * self.ndof * self.T, 2 * self.ndof + 3 * self.T
        self.vel = dq_lo is not None
        desc = _lib.oh_torque_desc(T=self.T, ndof=self.ndof, dt=float(dt), w_path=float(w_path), w_vel=float(w_vel), w_tau=float(w_tau),
                                   max_iter=int(max_iter), tol=float(tol), tol_compl=float(tol_compl), mu_barrier0=float(mu_barrier0), mu0=float(mu0),
                                   vel_limits=1 if self.vel else 0)
        lo = np.broadcast_to(np.asarray(-1e9 if tau_lo is None else tau_lo, dtype=np.float64), (self.ndof,))
        up = np.broadcast_to(np.asarray(1e9 if tau_up is None else tau_up, dtype=np.float64), (self.ndof,))
        for i in range(self.ndof):
            desc.tau_lo[i], desc.tau_up[i] = lo[i], up[i]
        if self.vel:
            vlo = np.broadcast_to(np.asarray(dq_lo, dtype=np.float64), (self.ndof,))
            vup = np.broadcast_to(np.asarray(dq_up, dtype=np.float64), (self.ndof,))
            for i in range(self.ndof):
                desc.dq_lo[i], desc.dq_up[i] = vlo[i], vup[i]
        self._h = C.c_void_p()
        _lib.check(lib.oh_create_torque(C.byref(desc), C.byref(self._h)), "oh_create_torque")
        _lib.check(lib.oh_set_constants(self._h, C.byref(chain)), "oh_set_constants")
        _lib.check(lib.oh_set_dynamics(self._h, C.byref(dynamics)), "oh_set_dynamics")
        self.chain, self.dynamics = chain, dynamics

    def multipliers(self, B: int):
        """(B, T, 2 ndof): multipliers >= 0 of (TAU - lo, up - TAU) per knot of the last solve; with velocity limits (B, T, 4 ndof): those of
        (dQ - dq_lo, dq_up - dQ) follow."""
        out = np.empty((B, self.T, (4 if self.vel else 2) * self.ndof))
        _lib.check(_lib.load().oh_get_multipliers(self._h, int(B), _lib._ptr(out)), "oh_get_multipliers")
        return out

    def rollout(self, state0, goal_table, n_ticks: int, advance: int = 1, mu_warm: float = 1e-6):
        """Closed-loop receding horizon on the device (oh_tq_rollout; the reference's pattern, example/point_mass_mpc.py:156-175: seed = the previous
        solution).  state0 (B, 2 ndof) = (q, dq); goal_table (B, n_ticks * advance + T, 3): tick k tracks rows k * advance .. k * advance + T - 1.
        Returns states (n_ticks + 1, B, 2 ndof), tau0 (n_ticks, B, ndof), f, iters, status (n_ticks, B)."""
        state0 = _lib.as_f64(state0).reshape(-1, 2 * self.ndof)
        B = state0.shape[0]
        goal_table = _lib.as_f64(goal_table)
        assert goal_table.shape == (B, n_ticks * advance + self.T, 3), goal_table.shape
        states = np.empty((n_ticks + 1, B, 2 * self.ndof))
        tau0 = np.empty((n_ticks, B, self.ndof))
        f = np.empty((n_ticks, B))
        iters = np.empty((n_ticks, B), dtype=np.int32)
        status = np.empty((n_ticks, B), dtype=np.int32)
        _lib.check(_lib.load().oh_tq_rollout(self._h, B, int(n_ticks), int(advance), float(mu_warm), _lib._ptr(state0), _lib._ptr(goal_table), _lib._ptr(states),
                                             _lib._ptr(tau0), _lib._ptr(f), _lib._ptr(iters), _lib._ptr(status)), "oh_tq_rollout")
        return states, tau0, f, iters, status

    def timing(self) -> dict:
        out = (C.c_double * 11)()
        _lib.check(_lib.load().oh_get_timing(self._h, out), "oh_get_timing")
        return {"solve_ms": out[4], "iterations_launched": int(out[5]), "work_instances": out[6], "eval_ms": out[0], "step_ms": out[2]}  # (eval / step: profiled solves only)


class FigureEightBackend(_OptionsMixin):
    """OH_PROBLEM_FIGURE_EIGHT handle."""

    def __init__(
        self,
        chain: _lib.oh_chain,
        T: int,
        dt: float,
        local_path: np.ndarray,
        w_path: float = 1000.0,
        w_vel: float = 0.01,
        max_iter: int = 200,
        tol: float = 1e-6,
        tol_feas: float = 1e-9,
        hessian: int = _lib.OH_HESSIAN_HYBRID,
        mu0: float = 0.0,
        lock_orientation: bool = True,
        fix_dq0: bool = True,
        path_in_frame: bool = True,
        guards: "Optional[_lib.oh_guards]" = None,
        ndof: Optional[int] = None,
    ):
        """``chain=None`` (with ``ndof``) creates the handle without kinematic constants: a non-root rank of a multi-GPU job receives them with
        ``optas_amd.distributed.Communicator.broadcast_constants`` before its first solve."""
        lib = _lib.load()
        if chain is None:
            assert ndof is not None, "a handle without constants needs ndof"
            chain_arg, chain = None, _lib.oh_chain(ndof=int(ndof))
        else:
            chain_arg = chain
        self.T, self.ndof = int(T), int(chain.ndof)
        self.nx = self.ndof * self.T + self.ndof * (self.T - 1)
        self.np_ = self.ndof + (1 + self.T if chain.has_lead else 0)  # [qc_opt; lead angle of qc; lead angle per knot]
        lp = _lib.as_f64(local_path, (self.T, 3))
        self._lp = lp  # keep alive during oh_create
        desc = _lib.oh_problem_desc(
            kind=_lib.OH_PROBLEM_FIGURE_EIGHT,
            T=self.T,
            ndof=self.ndof,
            dt=float(dt),
            w_path=float(w_path),
            w_vel=float(w_vel),
            local_path=lp.ctypes.data_as(C.POINTER(C.c_double)),
            lock_orientation=1 if lock_orientation else 0,
            fix_dq0=1 if fix_dq0 else 0,
            path_in_frame=1 if path_in_frame else 0,
            max_iter=int(max_iter),
            tol=float(tol),
            tol_feas=float(tol_feas),
            hessian=int(hessian),
            mu0=float(mu0),
        )
        self._h = C.c_void_p()
        _lib.check(lib.oh_create(C.byref(desc), C.byref(self._h)), "oh_create")
        # largest batch of one oh_solve call, as the library states it (32-bit stage-array offsets of the sweep kernels, row pad included):
        # bigger ones go in chunks
        mb = C.c_int(0)
        _lib.check(lib.oh_max_batch(self._h, C.byref(mb)), "oh_max_batch")
        self.max_batch = (mb.value // 65536) * 65536 if mb.value >= 65536 else (mb.value or None)
        if chain_arg is not None:
            _lib.check(lib.oh_set_constants(self._h, C.byref(chain)), "oh_set_constants")
        self.chain = chain_arg
        self.guards = guards
        self.n_rows = 0
        if guards is not None:
            _lib.check(lib.oh_set_guards(self._h, C.byref(guards)), "oh_set_guards")
            self.np_ = self.ndof + guards.n_links + 4 * guards.n_obstacles
            self.n_rows = (2 * self.ndof if guards.limits else 0) + guards.n_links * guards.n_obstacles + (2 * self.ndof if guards.vel_limits else 0)

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def set_constants_device(self, dptr: int, nbytes: int) -> None:
        _lib.check(_lib.load().oh_set_constants_device(self._h, C.c_void_p(dptr), nbytes), "oh_set_constants_device")

    def constants(self) -> _lib.oh_chain:
        """The kinematic constants the handle holds (set locally or received by broadcast)."""
        out = _lib.oh_chain()
        _lib.check(_lib.load().oh_get_constants(self._h, C.byref(out)), "oh_get_constants")
        return out

    def solve(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        lib = _lib.load()
        x0 = _lib.as_f64(x0)
        p = _lib.as_f64(p)
        if x0.ndim == 1:
            x0 = x0.reshape(1, -1)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        B = x0.shape[0]
        assert x0.shape == (B, self.nx), f"x0 must be (B, {self.nx})"
        assert p.shape == (B, self.np_), f"p must be (B, {self.np_})"
        chunk = getattr(self, "max_batch", None)  # one oh_solve call of the locked trajectory family is bounded (optas_hip.h)
        if chunk and B > chunk:
            # the handle only remembers its last call: multipliers and timing of every chunk are collected here and served from the cache
            parts, lams, tms = [], [], []
            for i in range(0, B, chunk):
                parts.append(self.solve(x0[i : i + chunk], p[i : i + chunk]))
                lams.append(self.multipliers(len(parts[-1].f)))
                tms.append(self.timing())
            self._chunked = {"B": B, "lam": np.concatenate(lams), "timing": {k: type(tms[0][k])(sum(t[k] for t in tms)) for k in tms[0]}}
            return BatchResult(*(np.concatenate([getattr(r, k) for r in parts]) for k in ("x", "f", "kkt", "iters", "status")))
        self._chunked = None
        x = np.empty((B, self.nx))
        f = np.empty(B)
        kkt = np.empty((B, 3))
        iters = np.empty(B, dtype=np.int32)
        status = np.empty(B, dtype=np.int32)
        _lib.check(
            lib.oh_solve(self._h, B, _lib._ptr(x0), _lib._ptr(p), _lib._ptr(x), _lib._ptr(f), _lib._ptr(kkt), _lib._ptr(iters), _lib._ptr(status)),
            "oh_solve",
        )
        return BatchResult(x, f, kkt, iters, status)

    def solve_device(self, B: int, d_x0, d_p, d_x, d_f, d_kkt, d_iters, d_status) -> None:
        """All arguments are _lib.DeviceBuffer (or None for optional outputs)."""
        g = lambda b: None if b is None else b.ptr
        self._chunked = None
        _lib.check(
            _lib.load().oh_solve_device(self._h, int(B), g(d_x0), g(d_p), g(d_x), g(d_f), g(d_kkt), g(d_iters), g(d_status)),
            "oh_solve_device",
        )

    def multipliers(self, B: int) -> np.ndarray:
        """Orientation-locked family: (B, 4T) signed multipliers of the quaternion rows; with inequality rows (guards):
        (B, T, NC) multipliers >= 0 in the row order of oh_guards."""
        ch = getattr(self, "_chunked", None)
        if ch is not None:  # the last solve was split into several oh_solve calls
            if int(B) != ch["B"]:
                raise ValueError(f"multipliers({B}) after a chunked solve of {ch['B']} instances")
            return ch["lam"]
        lam = np.empty((B, self.T, self.n_rows)) if self.guards is not None else np.empty((B, 4 * self.T))
        _lib.check(_lib.load().oh_get_multipliers(self._h, int(B), _lib._ptr(lam)), "oh_get_multipliers")
        return lam

    def specialize(self) -> dict:
        """Compile (or fetch from the cache) and load the evaluation kernels specialised for this handle's chain (oh_specialize); the
        library does it by itself at the first solve of >= 4096 instances."""
        _lib.check(_lib.load().oh_specialize(self._h), "oh_specialize")
        return self.specialize_info()

    def specialize_info(self) -> dict:
        info = (C.c_double * 4)()
        _lib.check(_lib.load().oh_specialize_info(self._h, info), "oh_specialize_info")
        return {"loaded": bool(info[0]), "fk_jac_loaded": bool(info[1]), "seconds": info[2], "from_disk_cache": bool(info[3])}

    def kernel_info(self, name: str) -> dict:
        """Code-object facts of the kernel this handle launches under that name (the specialised one once loaded)."""
        out = (C.c_int * 5)()
        _lib.check(_lib.load().oh_kernel_info_handle(self._h, name.encode(), out), "oh_kernel_info_handle")
        v, sc, lds, blk, nb = list(out)
        return {"registers_per_lane": v, "scratch_bytes_per_lane": sc, "lds_bytes_per_block": lds, "block": blk, "blocks_per_cu": nb, "waves_per_simd": nb * blk / 64.0 / 4.0}

    def set_profiling(self, on: bool) -> None:
        _lib.check(_lib.load().oh_set_profiling(self._h, 1 if on else 0), "oh_set_profiling")

    def flag(self, name: str) -> int:
        """oh_get_flag: 'fuse_couple', 'tail_threshold', 'specialized'."""
        v = C.c_int(0)
        _lib.check(_lib.load().oh_get_flag(self._h, name.encode(), C.byref(v)), "oh_get_flag")
        return v.value

    def timing(self) -> dict:
        ch = getattr(self, "_chunked", None)
        if ch is not None:
            return dict(ch["timing"])
        out = (C.c_double * 11)()
        _lib.check(_lib.load().oh_get_timing(self._h, out), "oh_get_timing")
        return {
            "eval_ms": out[0],
            "eval_launches": int(out[1]),
            "step_ms": out[2],
            "step_launches": int(out[3]),
            "solve_ms": out[4],
            "iterations_launched": int(out[5]),
            "instance_launches": int(out[6]),
            "compactions": int(out[7]),
            "couple_ms": out[8],
            "rejected_steps": int(out[9]),
            "tail_iterations": int(out[10]),
        }

    def fk_jac_soa_device(self, n: int, d_q, d_pose, d_J) -> None:
        g = lambda b: None if b is None else b.ptr
        _lib.check(_lib.load().oh_fk_jac_soa_device(self._h, int(n), g(d_q), g(d_pose), g(d_J)), "oh_fk_jac_soa_device")

    def fk_jac_device(self, n: int, d_q, d_pose, d_J) -> None:
        g = lambda b: None if b is None else b.ptr
        _lib.check(_lib.load().oh_fk_jac_device(self._h, int(n), g(d_q), g(d_pose), g(d_J)), "oh_fk_jac_device")

    def event_timer_start(self) -> None:
        _lib.check(_lib.load().oh_event_timer_start(self._h), "oh_event_timer_start")

    def event_timer_stop(self) -> float:
        ms = C.c_double(0.0)
        _lib.check(_lib.load().oh_event_timer_stop(self._h, C.byref(ms)), "oh_event_timer_stop")
        return ms.value

    def close(self) -> None:
        if self._h:
            _lib.load().oh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _offsets(container) -> dict:
    """label -> first index in vec() order; containers of the reference (sx_container.py) have no offsets(): computed from the shapes."""
    if hasattr(container, "offsets"):
        return container.offsets()
    out, o = {}, 0
    for k, v in container.items():
        out[k] = o
        o += int(v.shape[0]) * int(v.shape[1])
    return out


class MultiArmBackend:
    """Separable multi-robot problem (example/dual_arm.py): one position-only tracking handle per arm; the arms of
    all B instances are solved as independent GPU instances and stitched back into the reference's x layout."""

    def __init__(self, spec, opt, max_iter=200, tol=1e-6, hessian=_lib.OH_HESSIAN_GAUSS_NEWTON):
        self.spec, self.opt = spec, opt
        self.nx, self.np_ = opt.nx, opt.np
        self.xoff, self.poff = _offsets(opt.decision_variables), _offsets(opt.parameters)
        self.arms = []
        for a in spec.arms:
            guards = None
            if a.guards is not None:
                gs = a.guards
                guards = _lib.oh_guards()
                guards.limits = 1 if gs.lo is not None else 0
                if gs.lo is not None:
                    for j in range(a.robot.ndof):
                        guards.q_lo[j], guards.q_up[j] = float(gs.lo[j]), float(gs.up[j])
                if gs.vlo is not None:  # joint-velocity rows (round 3: k_couple_free_vel)
                    guards.vel_limits = 1
                    for j in range(a.robot.ndof):
                        guards.dq_lo[j], guards.dq_up[j] = float(gs.vlo[j]), float(gs.vup[j])
                guards.n_links, guards.n_obstacles = len(gs.links), len(gs.obstacles)
                for l, (k, off) in enumerate(a.robot.link_attachments(a.link, gs.links)):
                    if k < 0:
                        raise NotImplementedError(f"sphere link '{gs.links[l]}' does not move with any joint of the chain to '{a.link}'")
                    guards.link_joint[l] = k
                    for i in range(3):
                        guards.link_offset[l][i] = float(off[i])
            be = FigureEightBackend(
                a.robot.kinematic_chain(a.link), spec.T, spec.dt, a.offsets, w_path=a.w_path, w_vel=a.w_vel, max_iter=max_iter, tol=tol,
                hessian=hessian, lock_orientation=False, fix_dq0=False, path_in_frame=False, guards=guards,
            )
            self.arms.append((a, be))

    def solve(self, x0: np.ndarray, p: np.ndarray) -> BatchResult:
        x0 = _lib.as_f64(x0).reshape(-1, self.nx)
        p = _lib.as_f64(p).reshape(-1, self.np_)
        B = x0.shape[0]
        x = np.empty((B, self.nx))
        f = np.zeros(B)
        kkt = np.zeros((B, 3))
        iters = np.zeros(B, dtype=np.int32)
        status = np.zeros(B, dtype=np.int32)
        for a, be in self.arms:
            n, T = be.ndof, be.T
            oq, odq, op = self.xoff[a.q_name], self.xoff[a.dq_name], self.poff[a.qc_name]
            xa = np.concatenate([x0[:, oq : oq + n * T], x0[:, odq : odq + n * (T - 1)]], axis=1)
            pa = p[:, op : op + n]
            if a.guards is not None:
                cols = [pa] + [p[:, [self.poff[lr]]] for lr in a.guards.link_radii]
                for pos, rad in a.guards.obstacles:
                    cols += [p[:, self.poff[pos] : self.poff[pos] + 3], p[:, [self.poff[rad]]]]
                pa = np.ascontiguousarray(np.concatenate(cols, axis=1))
            r = be.solve(xa, pa)
            x[:, oq : oq + n * T] = r.x[:, : n * T]
            x[:, odq : odq + n * (T - 1)] = r.x[:, n * T :]
            f += r.f
            kkt = np.maximum(kkt, r.kkt)
            iters = np.maximum(iters, r.iters)
            status = _lib.worse_status(status, r.status)
        return BatchResult(x, f, kkt, iters, status)

    def set_options(self, options=None, **kw):
        for _, be in self.arms:
            be.set_options(options, **kw)
        return self

    def solve_ms(self) -> float:
        """Device time of the last solve: the arms run one after the other on their own handles."""
        return float(sum(be.timing()["solve_ms"] for _, be in self.arms))

    def close(self) -> None:
        for _, be in self.arms:
            be.close()
