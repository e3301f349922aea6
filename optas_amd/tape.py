"""Expression trees -> one scalar instruction tape per problem: the stand-in for the CasADi SX tape the reference's NLP back-ends
interpret (optas/optimization.py:8-24, solver.py:333-398), for small dense problems that match none of the hand-written kernel
families.  The tape is evaluated *on the GPU only* (csrc/oh_tape.hip: one thread per instance runs the forward sweep and one reverse
sweep per output row); this module only builds it.

Instruction i writes register i (SSA).  ops:  CONST c | X k (decision variable k, vec() order) | P k (parameter k) | ADD a b |
SUB a b | MUL a b | DIV a b | NEG a | SIN a | COS a | ATAN2 a b | SQRT a | SQR a, and (round 4: what the reference's own graphs emit beyond
those -- Quaternion.getrpy, spatialmath.py:384-404, and optas.clip, __init__.py:29-41) ASIN a | FABS a | FMIN a b | FMAX a b | LT a b | LE a b |
EQ a b | NE a b | NOT a | AND a b | OR a b (1.0 / 0.0 valued, zero derivative) | IFZ a b (casadi's if_else_zero: b where a != 0, else 0;
if_else(c, x, y) = IFZ(c, x) + IFZ(NOT c, y)), and (round 5: user costs written with `from casadi import *`, optas/__init__.py:2) EXP a | LOG a, through which the
builder also expresses pow, tanh, sinh, cosh, acos, atan, asinh, acosh, atanh, log1p, expm1 and sign.  Common sub-expressions are shared (hash-consing), constants are folded.  Link functions (position / rotation / quaternion / geometric Jacobian of a serial chain) are expanded with the
same chain walk as RobotModel.get_global_link_transform (models.py:826-868) over scalar registers.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .builder import IntegrationResidual
from .expr import (Add, Atan2, Block, Const, Expr, Gather, LinkFunction, MatMul, Mul, ParamCol, ParamRef, PathInFrame, RobotStates, Rows, Scale, Square, StateCols,
                   StateRef, Sub, SumSqr, VCat, VarRef)
from .spatialmath import rpy2r

OP_CONST, OP_X, OP_P, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SIN, OP_COS, OP_ATAN2, OP_SQRT, OP_SQR = range(13)
OP_ASIN, OP_FABS, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_NOT, OP_AND, OP_OR, OP_IFZ = range(13, 25)
OP_EXP, OP_LOG = 25, 26  # round 5
N_OPS = 27
MAX_TAPE = 1 << 18


class UnsupportedInstruction(NotImplementedError):
    pass


class TapeBuilder:
    def __init__(self):
        self.op: List[int] = []
        self.a: List[int] = []
        self.b: List[int] = []
        self.c: List[float] = []
        self._memo: Dict[tuple, int] = {}
        self._const: Dict[int, float] = {}  # register -> value for CONST registers (folding)

    def _emit(self, op, a=0, b=0, c=0.0) -> int:
        key = (op, a, b, c)
        r = self._memo.get(key)
        if r is None:
            r = len(self.op)
            self.op.append(op)
            self.a.append(a)
            self.b.append(b)
            self.c.append(float(c))
            self._memo[key] = r
            if op == OP_CONST:
                self._const[r] = float(c)
        return r

    def const(self, v: float) -> int:
        return self._emit(OP_CONST, 0, 0, float(v) + 0.0)  # +0.0 folds -0.0 into 0.0

    def x(self, k: int) -> int:
        return self._emit(OP_X, k)

    def p(self, k: int) -> int:
        return self._emit(OP_P, k)

    def is_const(self, r: int, v=None) -> bool:
        return r in self._const and (v is None or self._const[r] == v)

    def add(self, a, b):
        if self.is_const(a, 0.0):
            return b
        if self.is_const(b, 0.0):
            return a
        if self.is_const(a) and self.is_const(b):
            return self.const(self._const[a] + self._const[b])
        return self._emit(OP_ADD, min(a, b), max(a, b))

    def sub(self, a, b):
        if self.is_const(b, 0.0):
            return a
        if self.is_const(a) and self.is_const(b):
            return self.const(self._const[a] - self._const[b])
        if self.is_const(a, 0.0):
            return self.neg(b)
        return self._emit(OP_SUB, a, b)

    def mul(self, a, b):
        if self.is_const(a, 0.0) or self.is_const(b, 0.0):
            return self.const(0.0)
        if self.is_const(a, 1.0):
            return b
        if self.is_const(b, 1.0):
            return a
        if self.is_const(a) and self.is_const(b):
            return self.const(self._const[a] * self._const[b])
        if a == b:
            return self._emit(OP_SQR, a)
        return self._emit(OP_MUL, min(a, b), max(a, b))

    def div(self, a, b):
        if self.is_const(b, 1.0):
            return a
        if self.is_const(a) and self.is_const(b) and self._const[b] != 0.0:
            return self.const(self._const[a] / self._const[b])
        return self._emit(OP_DIV, a, b)

    def sqrt(self, a):
        return self.const(np.sqrt(self._const[a])) if self.is_const(a) and self._const[a] >= 0.0 else self._emit(OP_SQRT, a)

    def neg(self, a):
        if self.is_const(a):
            return self.const(-self._const[a])
        return self._emit(OP_NEG, a)

    def sin(self, a):
        return self.const(np.sin(self._const[a])) if self.is_const(a) else self._emit(OP_SIN, a)

    def cos(self, a):
        return self.const(np.cos(self._const[a])) if self.is_const(a) else self._emit(OP_COS, a)

    def atan2(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(np.arctan2(self._const[a], self._const[b]))
        return self._emit(OP_ATAN2, a, b)

    def sqr(self, a):
        return self.mul(a, a)

    # ---- round 4: the rest of what the reference's graphs emit (values as casadi's SX virtual machine computes them) -------------------------
    def _fold1(self, op, a, fn):
        return self.const(fn(self._const[a])) if self.is_const(a) else self._emit(op, a)

    def _fold2(self, op, a, b, fn):
        return self.const(fn(self._const[a], self._const[b])) if self.is_const(a) and self.is_const(b) else self._emit(op, a, b)

    def asin(self, a):
        return self._fold1(OP_ASIN, a, lambda v: float(np.arcsin(v)))

    def fabs(self, a):
        return self._fold1(OP_FABS, a, abs)

    def fmin(self, a, b):
        return self._fold2(OP_FMIN, a, b, min)

    def fmax(self, a, b):
        return self._fold2(OP_FMAX, a, b, max)

    def lt(self, a, b):
        return self._fold2(OP_LT, a, b, lambda x, y: float(x < y))

    def le(self, a, b):
        return self._fold2(OP_LE, a, b, lambda x, y: float(x <= y))

    def eq(self, a, b):
        return self._fold2(OP_EQ, a, b, lambda x, y: float(x == y))

    def ne(self, a, b):
        return self._fold2(OP_NE, a, b, lambda x, y: float(x != y))

    def lnot(self, a):
        return self._fold1(OP_NOT, a, lambda v: float(v == 0.0))

    def land(self, a, b):
        return self._fold2(OP_AND, a, b, lambda x, y: float(x != 0.0 and y != 0.0))

    def lor(self, a, b):
        return self._fold2(OP_OR, a, b, lambda x, y: float(x != 0.0 or y != 0.0))

    def ifz(self, c, x):
        """casadi's if_else_zero(c, x): x where c != 0, else 0."""
        if self.is_const(c):
            return x if self._const[c] != 0.0 else self.const(0.0)
        return self._emit(OP_IFZ, c, x)

    # ---- round 5: exp and log as instructions; the other elementary functions casadi offers (`from casadi import *`, optas/__init__.py:2)
    # composed from the instruction set, so that every evaluator, the reverse sweeps included, has two cases more and not twelve ---------------
    def exp(self, a):
        return self._fold1(OP_EXP, a, lambda v: float(np.exp(v)))

    def log(self, a):
        return self._fold1(OP_LOG, a, lambda v: float(np.log(v)) if v > 0.0 else (float("-inf") if v == 0.0 else float("nan")))

    def pow(self, a, b):
        """x ** y: repeated squaring for every constant integer exponent of magnitude up to 64 (finite for x <= 0 like casadi's OP_POW / OP_CONSTPOW: x**9 of a
        negative x is a number, not exp(9 log x) = NaN -- ADVICE r5), sqrt / reciprocal, otherwise exp(y log x) -- defined for x > 0, where its value and both partial
        derivatives are casadi's (y x^(y-1), x^y log x).  A constant integer exponent beyond 64 is refused rather than lowered to something that is wrong for x <= 0."""
        if self.is_const(b):
            e = self._const[b]
            if e == 0.5:
                return self.sqrt(a)
            if e == -1.0:
                return self.div(self.const(1.0), a)
            if np.isfinite(e) and e == int(e):
                k = int(e)
                if abs(k) > 64:
                    raise UnsupportedInstruction(f"pow with the constant integer exponent {k}: beyond the squaring chain (|k| <= 64), and exp(k log x) is wrong for x <= 0")
                if k < 0:
                    return self.div(self.const(1.0), self.pow(a, self.const(float(-k))))
                out, base = self.const(1.0), a
                while k:
                    if k & 1:
                        out = self.mul(out, base)
                    k >>= 1
                    if k:
                        base = self.sqr(base)
                return out
        return self.exp(self.mul(b, self.log(a)))

    def tanh(self, a):
        """1 - 2 / (exp(2x) + 1): exact limits -1 / +1 where exp under- / overflows."""
        return self.sub(self.const(1.0), self.div(self.const(2.0), self.add(self.exp(self.add(a, a)), self.const(1.0))))

    def sinh(self, a):
        return self.mul(self.const(0.5), self.sub(self.exp(a), self.exp(self.neg(a))))

    def cosh(self, a):
        return self.mul(self.const(0.5), self.add(self.exp(a), self.exp(self.neg(a))))

    def acos(self, a):
        return self.sub(self.const(float(np.pi / 2)), self.asin(a))

    def atan(self, a):
        return self.atan2(a, self.const(1.0))

    def tan(self, a):
        return self.div(self.sin(a), self.cos(a))

    def asinh(self, a):
        return self.log(self.add(a, self.sqrt(self.add(self.sqr(a), self.const(1.0)))))

    def acosh(self, a):
        return self.log(self.add(a, self.sqrt(self.sub(self.sqr(a), self.const(1.0)))))

    def atanh(self, a):
        return self.mul(self.const(0.5), self.log(self.div(self.add(self.const(1.0), a), self.sub(self.const(1.0), a))))

    def log1p(self, a):
        return self.log(self.add(self.const(1.0), a))

    def expm1(self, a):
        return self.sub(self.exp(a), self.const(1.0))

    def sign(self, a):
        """(x > 0) - (x < 0); zero derivative like casadi's."""
        z = self.const(0.0)
        return self.sub(self.lt(z, a), self.lt(a, z))

    def if_else(self, c, x, y):
        """casadi.if_else(c, x, y) as the SX graph holds it: if_else_zero(c, x) + if_else_zero(!c, y)."""
        return self.add(self.ifz(c, x), self.ifz(self.lnot(c), y))

    # ---- small dense helpers over arrays of registers --------------------------------------------------------------
    def mat_const(self, M) -> np.ndarray:
        M = np.atleast_2d(np.asarray(M, dtype=np.float64))
        return np.array([[self.const(v) for v in row] for row in M], dtype=np.int64)

    def matmul(self, A: np.ndarray, B: np.ndarray) -> np.ndarray:
        out = np.zeros((A.shape[0], B.shape[1]), dtype=np.int64)
        for i in range(A.shape[0]):
            for j in range(B.shape[1]):
                acc = self.const(0.0)
                for k in range(A.shape[1]):
                    acc = self.add(acc, self.mul(int(A[i, k]), int(B[k, j])))
                out[i, j] = acc
        return out

    def madd(self, A, B):
        A, B = np.broadcast_arrays(A, B)
        return np.array([[self.add(int(a), int(b)) for a, b in zip(ra, rb)] for ra, rb in zip(A, B)], dtype=np.int64)

    def msub(self, A, B):
        A, B = np.broadcast_arrays(A, B)
        return np.array([[self.sub(int(a), int(b)) for a, b in zip(ra, rb)] for ra, rb in zip(A, B)], dtype=np.int64)


def _rot_axis(tb: TapeBuilder, axis, th: int) -> np.ndarray:
    """Rodrigues rotation about the constant unit axis by the register angle th: I + sin K + (1 - cos) K^2 (spatialmath.py:89-99)."""
    a = np.asarray(axis, dtype=np.float64)
    K = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
    K2 = K @ K
    s, c = tb.sin(th), tb.cos(th)
    omc = tb.sub(tb.const(1.0), c)
    R = np.zeros((3, 3), dtype=np.int64)
    for i in range(3):
        for j in range(3):
            v = tb.const(1.0 if i == j else 0.0)
            v = tb.add(v, tb.mul(tb.const(K[i, j]), s))
            v = tb.add(v, tb.mul(tb.const(K2[i, j]), omc))
            R[i, j] = v
    return R


def _chain_walk(tb: TapeBuilder, robot, link: str, q: np.ndarray):
    """Symbolic models.py:826-868 over registers: returns (R 3x3, p 3x1, list of (axis_world 3, origin_world 3, jtype, actuated index))."""
    root = robot.urdf.get_root()
    R, p = tb.mat_const(np.eye(3)), tb.mat_const(np.zeros((3, 1)))
    joints = []
    names = robot.urdf.get_chain(root, link, links=False) if link != root else []
    for name in names:
        joint = robot.urdf.joint_map[name]
        xyz, rpy = robot.get_joint_origin(joint)
        p = tb.madd(p, tb.matmul(R, tb.mat_const(np.asarray(xyz).reshape(3, 1))))
        R = tb.matmul(R, tb.mat_const(rpy2r(rpy)))
        if joint.type == "fixed":
            continue
        idx = robot.get_actuated_joint_index(joint.name)
        axis = robot.get_joint_axis(joint)
        zw = tb.matmul(R, tb.mat_const(np.asarray(axis).reshape(3, 1)))
        if joint.type in ("revolute", "continuous"):
            joints.append((zw, p.copy(), 0, idx))
            R = tb.matmul(R, _rot_axis(tb, axis, int(q[idx])))
        elif joint.type == "prismatic":
            joints.append((zw, p.copy(), 1, idx))
            p = tb.madd(p, np.array([[tb.mul(int(zw[i, 0]), int(q[idx]))] for i in range(3)], dtype=np.int64))
        else:
            raise NotImplementedError(f"{joint.type} joints are currently not supported")
    return R, p, joints


def _chain_quat(tb: TapeBuilder, robot, link: str, q: np.ndarray) -> np.ndarray:
    """models.py:1049-1088 over registers: the xyzw quaternion of the link as the reference accumulates it, quat <- fromrpy(rpy) * quat, then
    quat <- fromangvec(q_i, axis) * quat per joint, with the class's own (reversed) product (spatialmath.py:298-312) -- its sign is part of what
    an equality row on a quaternion pins, so it is not rebuilt from the rotation matrix.  Returns 4 x 1 registers."""
    from .spatialmath import Quaternion

    def mul(a, b):  # Quaternion.__mul__: self = a, quat = b (spatialmath.py:298-312)
        x0, y0, z0, w0 = a
        x1, y1, z1, w1 = b
        m, ad, sb, ng = tb.mul, tb.add, tb.sub, tb.neg
        return (ad(sb(ad(m(x1, w0), m(y1, z0)), m(z1, y0)), m(w1, x0)),
                ad(ad(ad(ng(m(x1, z0)), m(y1, w0)), m(z1, x0)), m(w1, y0)),
                ad(ad(sb(m(x1, y0), m(y1, x0)), m(z1, w0)), m(w1, z0)),
                ad(sb(sb(ng(m(x1, x0)), m(y1, y0)), m(z1, z0)), m(w1, w0)))

    quat = tuple(tb.const(v) for v in (0.0, 0.0, 0.0, 1.0))
    root = robot.urdf.get_root()
    names = robot.urdf.get_chain(root, link, links=False) if link != root else []
    for name in names:
        joint = robot.urdf.joint_map[name]
        _, rpy = robot.get_joint_origin(joint)
        quat = mul(tuple(tb.const(float(v)) for v in Quaternion.fromrpy(rpy).getquat()), quat)
        if joint.type in ("fixed", "prismatic"):
            continue
        if joint.type not in ("revolute", "continuous"):
            raise NotImplementedError(f"{joint.type} joints are currently not supported")
        th = int(q[robot.get_actuated_joint_index(joint.name)])
        half = tb.mul(tb.const(0.5), th)
        s, c = tb.sin(half), tb.cos(half)
        ax = np.asarray(robot.get_joint_axis(joint), dtype=np.float64)
        ax = ax / np.linalg.norm(ax)
        quat = mul((tb.mul(tb.const(ax[0]), s), tb.mul(tb.const(ax[1]), s), tb.mul(tb.const(ax[2]), s), c), quat)
    return np.array([[int(v)] for v in quat], dtype=np.int64)


class Compiler:
    def __init__(self, opt):
        self.opt = opt
        self.tb = TapeBuilder()
        self.xoff = opt.decision_variables.offsets()
        self.poff = opt.parameters.offsets()
        self._cache: Dict[int, np.ndarray] = {}

    def block(self, container_off: Dict[str, int], label: str, shape, kind: str) -> np.ndarray:
        m, n = shape
        off = container_off[label]
        mk = self.tb.x if kind == "x" else self.tb.p
        return np.array([[mk(off + j * m + i) for j in range(n)] for i in range(m)], dtype=np.int64)  # column-major blocks

    def compile(self, e: Expr) -> np.ndarray:
        key = id(e)
        if key not in self._cache:
            self._cache[key] = self._compile(e)
        return self._cache[key]

    def _compile(self, e: Expr) -> np.ndarray:
        tb = self.tb
        if isinstance(e, Const):
            return tb.mat_const(e.value)
        if isinstance(e, ParamRef):
            return self.block(self.poff, e.name, e.shape, "p")
        if isinstance(e, ParamCol):
            return self.block(self.poff, e.param.name, e.param.shape, "p")[:, [e.col]]
        if isinstance(e, StateRef):
            full = self.block(self.xoff, e.var_name, (e.m, e.n), "x")
            return full if e.t is None else full[:, [e.t]]
        if isinstance(e, StateCols):
            return self.block(self.xoff, e.state.var_name, (e.state.m, e.state.n), "x")[:, e.lo : e.hi]
        if isinstance(e, VarRef):
            return self.block(self.xoff, e.var_name, e.shape, "x")
        if isinstance(e, RobotStates):
            X, Pm = self.compile(e.states), self.compile(e.params)
            full = np.zeros(e.shape, dtype=np.int64)
            full[list(e.opt_idx), :] = X
            full[list(e.par_idx), :] = Pm
            return full
        if isinstance(e, Rows):
            return self.compile(e.a)[list(e.idx), :]
        if isinstance(e, Block):
            return self.compile(e.a)[np.ix_(list(e.ridx), list(e.cidx))]
        if isinstance(e, VCat):
            return np.vstack([np.broadcast_to(self.compile(q), q.shape) for q in e.parts])
        if isinstance(e, Gather):
            va = np.broadcast_to(self.compile(e.a), e.a.shape).T.reshape(-1)
            out = np.empty(e.shape, dtype=np.int64)
            for (r, c), k in np.ndenumerate(e.idx):
                sg = float(e.sign[r, c])
                out[r, c] = tb.const(0.0) if k < 0 or sg == 0.0 else (int(va[k]) if sg == 1.0 else tb.mul(tb.const(sg), int(va[k])))
            return out
        if isinstance(e, Sub):
            return tb.msub(self.compile(e.a), self.compile(e.b))
        if isinstance(e, Add):
            return tb.madd(self.compile(e.a), self.compile(e.b))
        if isinstance(e, Scale):
            w = tb.const(e.w)
            return np.array([[tb.mul(w, int(v)) for v in row] for row in self.compile(e.a)], dtype=np.int64)
        if isinstance(e, Mul):
            A, B = np.broadcast_arrays(self.compile(e.a), self.compile(e.b))
            return np.array([[tb.mul(int(a), int(b)) for a, b in zip(ra, rb)] for ra, rb in zip(A, B)], dtype=np.int64)
        if isinstance(e, MatMul):
            return tb.matmul(self.compile(e.a), self.compile(e.b))
        if isinstance(e, Square):
            return np.array([[tb.sqr(int(v)) for v in row] for row in self.compile(e.a)], dtype=np.int64)
        if isinstance(e, SumSqr):
            acc = tb.const(0.0)
            for v in self.compile(e.a).reshape(-1):
                acc = tb.add(acc, tb.sqr(int(v)))
            return np.array([[acc]], dtype=np.int64)
        if isinstance(e, Atan2):
            Y, X = np.broadcast_arrays(self.compile(e.y), self.compile(e.x))
            return np.array([[tb.atan2(int(a), int(b)) for a, b in zip(ra, rb)] for ra, rb in zip(Y, X)], dtype=np.int64)
        if isinstance(e, IntegrationResidual):
            X, Xd = self.compile(e.x), self.compile(e.xd)[:, : e.n]
            dt = np.array([[tb.const(v) for v in e.dt]], dtype=np.int64)
            step = np.array([[tb.mul(int(dt[0, j]), int(Xd[i, j])) for j in range(e.n)] for i in range(Xd.shape[0])], dtype=np.int64)
            return tb.msub(tb.madd(X[:, :-1][:, : e.n], step), X[:, 1:][:, : e.n])
        if isinstance(e, PathInFrame):
            o, R = self.compile(e.origin), self.compile(e.rotation)
            return tb.madd(np.broadcast_to(o.reshape(3, 1), (3, e.local.shape[1])), tb.matmul(R, tb.mat_const(e.local)))
        if isinstance(e, LinkFunction):
            Q = self.compile(e.q)
            cols = []
            for j in range(Q.shape[1]):
                R, p, joints = _chain_walk(tb, e.robot, e.link, Q[:, j])
                if e.what == "position":
                    cols.append(p)
                elif e.what == "rotation":
                    return R
                elif e.what == "geometric_jacobian":
                    J = np.full((6, e.robot.ndof), tb.const(0.0), dtype=np.int64)
                    for zw, pj, jt, idx in joints:
                        if jt == 0:
                            d = tb.msub(p, pj)
                            cr = [tb.sub(tb.mul(int(zw[1, 0]), int(d[2, 0])), tb.mul(int(zw[2, 0]), int(d[1, 0]))),
                                  tb.sub(tb.mul(int(zw[2, 0]), int(d[0, 0])), tb.mul(int(zw[0, 0]), int(d[2, 0]))),
                                  tb.sub(tb.mul(int(zw[0, 0]), int(d[1, 0])), tb.mul(int(zw[1, 0]), int(d[0, 0])))]
                            for i in range(3):
                                J[i, idx] = cr[i]
                                J[3 + i, idx] = zw[i, 0]
                        else:
                            for i in range(3):
                                J[i, idx] = zw[i, 0]
                    return J
                else:
                    cols.append(_chain_quat(tb, e.robot, e.link, Q[:, j]))
            return np.hstack(cols)
        raise NotImplementedError(f"cannot compile {type(e).__name__} to a tape")


@dataclass
class Tape:
    op: np.ndarray  # int32 [L]
    a: np.ndarray  # int32 [L]
    b: np.ndarray  # int32 [L]
    c: np.ndarray  # float64 [L]
    out_cost: int  # register of f
    out_rows: np.ndarray  # int32 [nrows] registers of the constraint rows in the order k, g, a, h (rows >= 0 first, then == 0)
    n_ineq: int  # k and g rows (>= 0)
    n_eq: int  # a and h rows (== 0)
    nx: int
    np_: int


def compile_problem(opt) -> Tape:
    """One tape for f and every row of k, g (>= 0) and a, h (= 0) of the Optimization (the rows of v without the mirrored -a, -h)."""
    comp = Compiler(opt)
    tb = comp.tb
    f = tb.const(0.0)
    for term in opt.cost_terms.values():
        f = tb.add(f, int(comp.compile(term).reshape(-1)[0]))
    rows: List[int] = []

    def vec(container):
        out = []
        for term in container.values():
            m, n = term.shape
            out += [int(v) for v in np.broadcast_to(comp.compile(term), (m, n)).T.reshape(-1)]
        return out

    ineq = vec(opt.lin_ineq_constraints) + vec(opt.ineq_constraints)
    eq = vec(opt.lin_eq_constraints) + vec(opt.eq_constraints)
    rows = ineq + eq
    if len(tb.op) > MAX_TAPE:
        raise NotImplementedError(f"tape of {len(tb.op)} instructions exceeds {MAX_TAPE}")
    return Tape(np.asarray(tb.op, dtype=np.int32), np.asarray(tb.a, dtype=np.int32), np.asarray(tb.b, dtype=np.int32), np.asarray(tb.c, dtype=np.float64),
                int(f), np.asarray(rows, dtype=np.int32), len(ineq), len(eq), opt.nx, opt.np)


_BINARY_NONLINEAR = (OP_ATAN2, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_AND, OP_OR, OP_IFZ)


def tape_degrees(tape: Tape) -> np.ndarray:
    """Degree in x of every register of a tape: 0 constant / parameter only, 1 affine, 2 quadratic, 3 anything else -- the classification the
    reference asks CasADi for (cs.is_linear / is_quadratic, builder.py:226-240), here on the instruction arrays."""
    deg = np.zeros(len(tape.op), dtype=np.int8)
    for r, (op, a, b) in enumerate(zip(tape.op, tape.a, tape.b)):
        if op == OP_X:
            deg[r] = 1
        elif op in (OP_CONST, OP_P):
            deg[r] = 0
        elif op in (OP_ADD, OP_SUB):
            deg[r] = max(deg[a], deg[b])
        elif op == OP_NEG:
            deg[r] = deg[a]
        elif op == OP_MUL:
            deg[r] = min(3, deg[a] + deg[b])
        elif op == OP_SQR:
            deg[r] = min(3, 2 * deg[a])
        elif op == OP_DIV:
            deg[r] = deg[a] if deg[b] == 0 else 3
        else:  # sin, cos, atan2, sqrt
            deg[r] = 0 if max(deg[a], deg[b] if op in _BINARY_NONLINEAR else 0) == 0 else 3
    return deg


def rebalance_sums(tape: Tape) -> Tape:
    """The same problem with every long chain of additions / subtractions re-associated into a balanced tree.  Front ends write a sum of k
    terms (sumsqr over a trajectory, a row of a matrix product) as a chain of k - 1 dependent ADDs: for the one-thread-per-instance evaluators the
    order is irrelevant, for the wavefront-per-instance evaluator (csrc/oh_tape_wave.hip), which executes a dependency level per pass, the chain *is*
    the critical path (example/simple_joint_space_planner.py: 143 levels, 126 of them fewer than 8 operations wide; 27 after this).  A register is
    interior to a sum when it is an ADD / SUB consumed exactly once, by another ADD / SUB, and is not an output; every maximal tree of such
    registers is flattened into signed leaves and summed pairwise.  Values change by the rounding of the summation order only (|d| <= 1e-13 on the
    planner's rows); dead registers are dropped on the way."""
    op, ra, rb = tape.op, tape.a, tape.b
    L = len(op)
    binary = {OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_ATAN2, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_AND, OP_OR, OP_IFZ}
    outs = set(int(r) for r in tape.out_rows) | {int(tape.out_cost)}
    live = np.zeros(L, dtype=bool)
    live[list(outs)] = True
    n_cons = np.zeros(L, dtype=np.int64)
    sum_cons = np.zeros(L, dtype=np.int64)  # consumers that are ADD / SUB
    for i in range(L - 1, -1, -1):
        if not live[i] or op[i] < OP_ADD:
            continue
        ops_ = (int(ra[i]), int(rb[i])) if int(op[i]) in binary else (int(ra[i]),)
        for r in ops_:
            live[r] = True
            n_cons[r] += 1
            if op[i] in (OP_ADD, OP_SUB):
                sum_cons[r] += 1
    is_sum = (op == OP_ADD) | (op == OP_SUB)
    interior = is_sum & live & (n_cons == 1) & (sum_cons == 1)
    for r in outs:
        interior[r] = False
    ops, aa, bb, cc = [], [], [], []
    new = np.full(L, -1, dtype=np.int64)

    def emit(o, a=0, b=0, c=0.0):
        ops.append(o), aa.append(a), bb.append(b), cc.append(c)
        return len(ops) - 1

    for i in range(L):
        if not live[i] or interior[i]:
            continue
        o = int(op[i])
        if not is_sum[i]:
            new[i] = emit(o, int(new[ra[i]]) if o >= OP_ADD else int(ra[i]), int(new[rb[i]]) if o in binary else 0, float(tape.c[i]))
            continue
        leaves, stack = [], [(i, 1)]
        while stack:  # depth first, left operand first: the leaves in the order the chain adds them
            r, sg = stack.pop()
            if r == i or interior[r]:
                stack.append((int(rb[r]), sg if op[r] == OP_ADD else -sg))
                stack.append((int(ra[r]), sg))
            else:
                leaves.append((int(new[r]), sg))
        while len(leaves) > 1:
            nxt = []
            for k in range(0, len(leaves) - 1, 2):
                (u, su), (w, sw) = leaves[k], leaves[k + 1]
                if su == sw:
                    nxt.append((emit(OP_ADD, u, w), su))
                elif su > 0:
                    nxt.append((emit(OP_SUB, u, w), 1))
                else:
                    nxt.append((emit(OP_SUB, w, u), 1))
            if len(leaves) % 2:
                nxt.append(leaves[-1])
            leaves = nxt
        r, sg = leaves[0]
        new[i] = r if sg > 0 else emit(OP_NEG, r)
    return Tape(np.asarray(ops, dtype=np.int32), np.asarray(aa, dtype=np.int32), np.asarray(bb, dtype=np.int32), np.asarray(cc, dtype=np.float64),
                int(new[tape.out_cost]), np.asarray([new[int(r)] for r in tape.out_rows], dtype=np.int32), tape.n_ineq, tape.n_eq, tape.nx, tape.np_)


def band_rewrite(tape: Tape) -> Optional[Tape]:
    """The tape of the equivalent linearly constrained QP when the cost is at most quadratic in x, the equality rows affine, and every
    inequality row either affine or of the form ``c - e*e`` with ``e`` affine and ``c`` a non-negative constant (example/torque_control_example.py:93-95):
    such a row is the band -sqrt(c) <= e <= sqrt(c) (the same feasible set, the same minimisers), written as the two rows e + sqrt(c), sqrt(c) - e.
    None when the problem is not of that shape.  The tape-level twin of lowering._band_rows, for problems that arrive as CasADi functions."""
    deg = tape_degrees(tape)
    if deg[tape.out_cost] > 2:
        return None
    ni = tape.n_ineq
    if any(deg[r] > 1 for r in tape.out_rows[ni:]):
        return None
    op, ra, rb, rc = tape.op, tape.a, tape.b, tape.c

    def square_of(r):  # register e with r = e*e, or None
        if op[r] == OP_SQR:
            return int(ra[r])
        if op[r] == OP_MUL and ra[r] == rb[r]:
            return int(ra[r])
        return None

    def band(r):  # (e, c) for r = c - e*e in the forms a front end writes it
        if op[r] == OP_SUB and op[ra[r]] == OP_CONST:
            return square_of(int(rb[r])), float(rc[ra[r]])
        if op[r] == OP_ADD:
            for u, w in ((ra[r], rb[r]), (rb[r], ra[r])):
                if op[u] == OP_CONST and op[w] == OP_NEG:
                    return square_of(int(ra[w])), float(rc[u])
        return None, 0.0

    ops, aa, bb, cc = list(op), list(ra), list(rb), list(rc)

    def emit(o, a=0, b=0, c=0.0):
        ops.append(o), aa.append(a), bb.append(b), cc.append(c)
        return len(ops) - 1

    rows, found = [], 0
    for r in tape.out_rows[:ni]:
        r = int(r)
        if deg[r] <= 1:
            rows.append(r)
            continue
        e, c = band(r)
        if e is None or deg[e] > 1 or not c >= 0.0:
            return None
        half = emit(OP_CONST, 0, 0, float(np.sqrt(c)))
        rows += [emit(OP_ADD, min(e, half), max(e, half)), emit(OP_SUB, half, e)]
        found += 1
    if not found:
        return None
    out_rows = np.asarray(rows + [int(r) for r in tape.out_rows[ni:]], dtype=np.int32)
    return Tape(np.asarray(ops, dtype=np.int32), np.asarray(aa, dtype=np.int32), np.asarray(bb, dtype=np.int32), np.asarray(cc, dtype=np.float64),
                tape.out_cost, out_rows, len(rows), tape.n_eq, tape.nx, tape.np_)


# ---- round 5: affine equality rows eliminated by substitution ------------------------------------------------------------------------
# A trajectory problem written with the reference's builder carries its dynamics as linear equality rows -- integrate_model_states, fix_configuration,
# initial_configuration (builder.py:419-469, 511-539): 147 of the 154 equality rows of example/simple_joint_space_planner.py.  The generic family
# solves by augmented Lagrangian + (L-)BFGS, and the penalty on those rows is what its quasi-Newton iteration crawls on (2600 evaluations).  They
# are affine in x with constant coefficients, so they can be satisfied identically: Gauss-Jordan on their coefficient matrix picks one pivot variable
# per row, x_pivot = -N x_free - M c(p), and the tape is re-emitted over the free variables (single shooting, done by the host once per handle).
# The planner: 280 -> 133 variables, 6281 -> 3363 instructions, 2643 -> 447 evaluations at the same optimum (numpy restatement of the solver).
_BINARY_OPS = frozenset({OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_ATAN2, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_AND, OP_OR, OP_IFZ})


def affine_forms(tape: Tape, regs) -> List[Optional[Dict[int, float]]]:
    """For each register in `regs`: {variable index: coefficient} if the register is affine in x with coefficients that are numeric constants (they
    may not depend on p: the elimination is done once per handle), else None.  The constant part (any function of p) is not represented."""
    deg = tape_degrees(tape)
    need = np.zeros(len(tape.op), dtype=bool)
    stack = [int(r) for r in regs if deg[int(r)] <= 1]
    while stack:
        r = stack.pop()
        if need[r]:
            continue
        need[r] = True
        o = int(tape.op[r])
        if o >= OP_ADD:
            stack.append(int(tape.a[r]))
            if o in _BINARY_OPS:
                stack.append(int(tape.b[r]))
    form: Dict[int, Optional[Dict[int, float]]] = {}
    cval: Dict[int, Optional[float]] = {}  # numeric value of registers that are constants (no x, no p)
    for r in np.flatnonzero(need):
        r = int(r)
        o, a, b = int(tape.op[r]), int(tape.a[r]), int(tape.b[r])
        if o == OP_CONST:
            form[r], cval[r] = {}, float(tape.c[r])
        elif o == OP_P:
            form[r], cval[r] = {}, None
        elif o == OP_X:
            form[r], cval[r] = {a: 1.0}, None
        elif deg[r] == 0:  # a function of parameters / constants only
            form[r] = {}
            va, vb = cval.get(a), (cval.get(b) if o in _BINARY_OPS else 0.0)
            cval[r] = None
            if va is not None and vb is not None:
                cval[r] = {OP_ADD: lambda: va + vb, OP_SUB: lambda: va - vb, OP_MUL: lambda: va * vb, OP_NEG: lambda: -va, OP_SQR: lambda: va * va}.get(o, lambda: None)()
                if o == OP_DIV and vb != 0.0:
                    cval[r] = va / vb
        else:  # degree 1
            cval[r] = None
            fa, fb = form.get(a), (form.get(b) if o in _BINARY_OPS else None)
            out: Optional[Dict[int, float]] = None
            if o in (OP_ADD, OP_SUB) and fa is not None and fb is not None:
                out = dict(fa)
                sgn = 1.0 if o == OP_ADD else -1.0
                for k, v in fb.items():
                    out[k] = out.get(k, 0.0) + sgn * v
            elif o == OP_NEG and fa is not None:
                out = {k: -v for k, v in fa.items()}
            elif o == OP_MUL and fa is not None and fb is not None:
                if deg[a] == 0 and cval.get(a) is not None:
                    out = {k: cval[a] * v for k, v in fb.items()}
                elif deg[b] == 0 and cval.get(b) is not None:
                    out = {k: cval[b] * v for k, v in fa.items()}
            elif o == OP_DIV and fa is not None and deg[b] == 0 and cval.get(b) not in (None, 0.0):
                out = {k: v / cval[b] for k, v in fa.items()}
            form[r] = out
    return [form.get(int(r)) if deg[int(r)] <= 1 else None for r in regs]


def quadratic_cost_hessian(tape: Tape) -> Optional[np.ndarray]:
    """The constant part of the Hessian of the cost register: the sum over the terms of the cost's sum tree that are quadratic in x with numeric
    coefficients -- squares and products of affine forms, scaled by constants (sumsqr costs on states, velocities, accelerations: the terms a
    trajectory problem is mostly made of; builder.py:226-240 classifies the same way through cs.is_quadratic).  Terms of higher degree, and quadratic
    terms whose coefficients depend on p, contribute nothing: the result is a lower bound of the curvature that is known before the first solve, not
    the Hessian of an arbitrary cost.  None when no such term exists."""
    deg = tape_degrees(tape)
    root = int(tape.out_cost)
    if deg[root] < 2:
        return None
    # walk the sum tree from the root with a weight per path (hash-consed sub-sums are visited once per use, as the sum uses them)
    leaves: List[Tuple[float, int]] = []  # (weight, register) of SQR / MUL nodes that are products of two affine forms
    stack: List[Tuple[float, int]] = [(1.0, root)]
    cache_c: Dict[int, Optional[float]] = {}

    def cval(r: int) -> Optional[float]:
        """numeric value of a register that depends on neither x nor p (None otherwise)"""
        if r in cache_c:
            return cache_c[r]
        o, a, b = int(tape.op[r]), int(tape.a[r]), int(tape.b[r])
        out: Optional[float] = None
        if o == OP_CONST:
            out = float(tape.c[r])
        elif o in (OP_ADD, OP_SUB, OP_MUL, OP_DIV):
            va, vb = cval(a), cval(b)
            if va is not None and vb is not None:
                out = va + vb if o == OP_ADD else va - vb if o == OP_SUB else va * vb if o == OP_MUL else (va / vb if vb != 0.0 else None)
        elif o == OP_NEG:
            va = cval(a)
            out = None if va is None else -va
        elif o == OP_SQR:
            va = cval(a)
            out = None if va is None else va * va
        cache_c[r] = out
        return out

    budget = 4 * len(tape.op) + 64  # (a DAG whose sums share sub-sums many times over could unfold without end: give up rather than hang)
    while stack:
        budget -= 1
        if budget < 0:
            return None
        w, r = stack.pop()
        if deg[r] < 2:
            continue
        o, a, b = int(tape.op[r]), int(tape.a[r]), int(tape.b[r])
        if o == OP_ADD:
            stack += [(w, a), (w, b)]
        elif o == OP_SUB:
            stack += [(w, a), (-w, b)]
        elif o == OP_NEG:
            stack.append((-w, a))
        elif o == OP_SQR and deg[a] == 1:
            leaves.append((w, r))
        elif o == OP_MUL:
            if deg[a] == 1 and deg[b] == 1:
                leaves.append((w, r))
            elif deg[a] == 0 and cval(a) is not None:
                stack.append((w * cval(a), b))
            elif deg[b] == 0 and cval(b) is not None:
                stack.append((w * cval(b), a))
        elif o == OP_DIV and deg[b] == 0 and cval(b) not in (None, 0.0):
            stack.append((w / cval(b), a))
        # anything else (degree 3, parameter-dependent scale): no contribution
    if not leaves:
        return None
    operands = sorted({int(tape.a[r]) for _, r in leaves} | {int(tape.b[r]) for _, r in leaves if int(tape.op[r]) == OP_MUL})
    forms = dict(zip(operands, affine_forms(tape, operands)))
    n = int(tape.nx)
    Q = np.zeros((n, n))
    any_term = False
    for w, r in leaves:
        fa = forms.get(int(tape.a[r]))
        fb = fa if int(tape.op[r]) == OP_SQR else forms.get(int(tape.b[r]))
        if not fa or not fb:  # (None: coefficients depend on p; empty: no x)
            continue
        ia, va = np.fromiter(fa.keys(), dtype=np.int64), np.fromiter(fa.values(), dtype=np.float64)
        ib, vb = np.fromiter(fb.keys(), dtype=np.int64), np.fromiter(fb.values(), dtype=np.float64)
        blk = w * np.outer(va, vb)  # d2 (a b) = grad a grad b^T + grad b grad a^T
        Q[np.ix_(ia, ib)] += blk
        Q[np.ix_(ib, ia)] += blk.T
        any_term = True
    return Q if any_term else None


def quadratic_cost_metric(tape: Tape, max_n: int = 1024) -> Optional[np.ndarray]:
    """Initial metric H0 [nx][nx] for the limited-memory iteration on this tape (oh_tape_set_metric): the inverse of the constant block of the cost's
    Hessian, shifted where it is singular or nearly so (variables no quadratic term touches get the curvature 1e-3 of the largest: a long first step
    the line search cuts, not a division by zero).  None when the cost has no such block, is not convex on it, or the matrix would not pay (nx > max_n:
    the product r = H0 q is nx^2 multiply-adds per iteration and instance)."""
    n = int(tape.nx)
    if n > max_n:
        return None
    Q = quadratic_cost_hessian(tape)
    if Q is None:
        return None
    Q = 0.5 * (Q + Q.T)
    lam = np.linalg.eigvalsh(Q)
    if not np.isfinite(lam).all() or lam[-1] <= 0.0 or lam[0] < -1e-9 * lam[-1]:
        return None
    shift = max(0.0, 1e-3 * lam[-1] - lam[0])
    H0 = np.linalg.inv(Q + shift * np.eye(n))
    return np.ascontiguousarray(0.5 * (H0 + H0.T))


def reemit(tb: TapeBuilder, tape: Tape, roots, xreg) -> Dict[int, int]:
    """Copy the sub-graphs of `roots` into the builder with variable k read from register xreg(k); returns {old register: new register}."""
    need = np.zeros(len(tape.op), dtype=bool)
    stack = [int(r) for r in roots]
    while stack:
        r = stack.pop()
        if need[r]:
            continue
        need[r] = True
        o = int(tape.op[r])
        if o >= OP_ADD:
            stack.append(int(tape.a[r]))
            if o in _BINARY_OPS:
                stack.append(int(tape.b[r]))
    two = {OP_ADD: tb.add, OP_SUB: tb.sub, OP_MUL: tb.mul, OP_DIV: tb.div, OP_ATAN2: tb.atan2, OP_FMIN: tb.fmin, OP_FMAX: tb.fmax, OP_LT: tb.lt, OP_LE: tb.le,
           OP_EQ: tb.eq, OP_NE: tb.ne, OP_AND: tb.land, OP_OR: tb.lor, OP_IFZ: tb.ifz}
    one = {OP_NEG: tb.neg, OP_SIN: tb.sin, OP_COS: tb.cos, OP_SQRT: tb.sqrt, OP_SQR: tb.sqr, OP_ASIN: tb.asin, OP_FABS: tb.fabs, OP_NOT: tb.lnot, OP_EXP: tb.exp,
           OP_LOG: tb.log}
    new: Dict[int, int] = {}
    for r in np.flatnonzero(need):
        r = int(r)
        o, a, b = int(tape.op[r]), int(tape.a[r]), int(tape.b[r])
        if o == OP_CONST:
            new[r] = tb.const(float(tape.c[r]))
        elif o == OP_X:
            new[r] = xreg(a)
        elif o == OP_P:
            new[r] = tb.p(a)
        elif o in two:
            new[r] = two[o](new[a], new[b])
        else:
            new[r] = one[o](new[a])
    return new


@dataclass
class Elimination:
    """What ties a tape whose affine equality rows have been eliminated to the problem it came from."""
    tape: Tape            # over the free variables; rows: every inequality row, then the equality rows that stayed
    free: np.ndarray      # original indices of its variables, in its order
    pivot: np.ndarray     # original indices of the eliminated variables
    def_regs: np.ndarray  # registers of the reduced tape holding the eliminated variables (as functions of the free ones and p)
    rows_out: np.ndarray  # positions, among the original equality rows, of the eliminated rows (pivot[i] was taken from rows_out[i])
    rows_kept: np.ndarray  # positions of the equality rows that stayed, in the reduced tape's order
    A_pivot: np.ndarray   # coefficients of the eliminated rows on the eliminated variables [m, m]: A_pivot^T nu = dL/dx_pivot gives their multipliers


def eliminate_affine_equalities(tape: Tape, min_rows: int = 1) -> Optional[Elimination]:
    """None when the tape has fewer than `min_rows` equality rows that are affine in x with constant coefficients."""
    n_i, n_e = int(tape.n_ineq), int(tape.n_eq)
    eq_regs = [int(r) for r in tape.out_rows[n_i : n_i + n_e]]
    forms = affine_forms(tape, eq_regs)
    cand = [i for i, f in enumerate(forms) if f]  # (an empty form is a row without x: nothing to pivot on)
    if len(cand) < min_rows:
        return None
    nx = int(tape.nx)
    R = np.zeros((len(cand), nx))
    for i, ri in enumerate(cand):
        for k, v in forms[ri].items():
            R[i, k] = v
    A = R.copy()
    M = np.eye(len(cand))
    pivots: List[int] = []
    rows_used: List[int] = []
    taken = np.zeros(nx, dtype=bool)
    for i in range(len(cand)):
        row = np.where(taken, 0.0, R[i])
        big = np.abs(row).max()
        if not big > 1e-9 * max(1.0, np.abs(A[i]).max()):
            continue  # the row depends on the rows before it: it stays a row of the problem
        c = int(np.flatnonzero(np.abs(row) >= 0.5 * big)[-1])  # a well-scaled pivot; among those the last variable (q_{t+1} of an Euler row, not q_t)
        s = R[i, c]
        R[i] /= s
        M[i] /= s
        for j in range(len(cand)):
            if j != i and R[j, c] != 0.0:
                f = R[j, c]
                R[j] -= f * R[i]
                M[j] -= f * M[i]
        taken[c] = True
        pivots.append(c)
        rows_used.append(i)
    if len(pivots) < min_rows:
        return None
    free = np.flatnonzero(~taken)
    if len(free) == 0:
        return None
    tb = TapeBuilder()
    xnew = {int(k): tb.x(j) for j, k in enumerate(free)}
    zero = tb.const(0.0)
    # constant parts c_r(p) of the eliminated rows: the rows with every variable at zero
    sel = [cand[i] for i in rows_used]
    cmap = reemit(tb, tape, [eq_regs[ri] for ri in sel], lambda k: zero)
    c_regs = {i: cmap[eq_regs[cand[i]]] for i in rows_used}
    defs: Dict[int, int] = {}
    for i, k in zip(rows_used, pivots):
        terms = [tb.mul(tb.const(-R[i, f]), xnew[int(f)]) for f in free if abs(R[i, f]) > 1e-14]
        terms += [tb.mul(tb.const(-M[i, j]), c_regs[j]) for j in rows_used if abs(M[i, j]) > 1e-14]
        acc = zero
        for t_ in terms:
            acc = tb.add(acc, t_)
        defs[k] = acc
    sel_set = set(sel)
    kept = [i for i in range(n_e) if i not in sel_set]
    roots = [int(tape.out_cost)] + [int(r) for r in tape.out_rows[:n_i]] + [eq_regs[i] for i in kept]
    full = reemit(tb, tape, roots, lambda k: xnew[k] if k in xnew else defs[k])
    rows = [full[int(r)] for r in tape.out_rows[:n_i]] + [full[eq_regs[i]] for i in kept]
    # the definitions must survive dead-code removal downstream: they are read back after the solve (oh_tape_probe)
    red = Tape(np.asarray(tb.op, dtype=np.int32), np.asarray(tb.a, dtype=np.int32), np.asarray(tb.b, dtype=np.int32), np.asarray(tb.c, dtype=np.float64),
               int(full[int(tape.out_cost)]), np.asarray(rows, dtype=np.int32), n_i, len(kept), len(free), int(tape.np_))
    if len(red.op) > MAX_TAPE:
        return None
    return Elimination(red, free.astype(np.int64), np.asarray(pivots, dtype=np.int64), np.asarray([defs[k] for k in pivots], dtype=np.int32),
                       np.asarray(sel, dtype=np.int64), np.asarray(kept, dtype=np.int64), A[np.ix_(rows_used, pivots)].copy())
