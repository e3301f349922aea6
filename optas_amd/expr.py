"""A deliberately tiny expression vocabulary standing in for the CasADi SX graphs that
``OptimizationBuilder`` / ``RobotModel`` emit in the reference (optas/builder.py, optas/models.py).

CasADi is not available, and a general symbolic engine is not the point: the HIP backend lowers a
*recognised* set of task terms to hand-written kernels.  These node types are exactly the ones the
BASELINE configs are written in; ``optas_amd.lowering`` pattern-matches trees of them.  Anything else
raises ``NotImplementedError`` at ``HIPSolver.setup`` -- never silently.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np


class Expr:
    """Base node.  ``shape`` is (rows, cols) like a CasADi matrix."""

    shape: Tuple[int, int] = (1, 1)
    __array_ufunc__ = None  # numpy arrays on the left of an operator defer to __rsub__ / __radd__ / __rmul__ / __rmatmul__

    # -- the handful of operators the example scripts use ------------------------------------------------
    def __sub__(self, other):
        return Sub(self, as_expr(other))

    def __rsub__(self, other):
        return Sub(as_expr(other), self)

    def __add__(self, other):
        return Add(self, as_expr(other))

    def __radd__(self, other):
        return Add(as_expr(other), self)

    def __mul__(self, other):
        if isinstance(other, (int, float)):
            return Scale(float(other), self)
        if isinstance(other, Expr):
            return Mul(self, other)
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, (int, float)):
            return Scale(1.0 / float(other), self)
        return NotImplemented

    def __neg__(self):
        return Scale(-1.0, self)

    def __pow__(self, k):
        if k == 2:
            return Square(self)
        return NotImplemented

    def __matmul__(self, other):
        return _mtimes(self, other)

    def __rmatmul__(self, other):
        return _mtimes(other, self)

    @property
    def T(self):
        """Transpose (``Rc.T``, ``diffp.T @ W_p @ diffp`` in example/torque_control_example.py:69-85)."""
        return transpose(self)

    def __getitem__(self, key):
        """Row selection of a generic node: e[i], e[a:b] (``veff[:3]``, ``pn[2]`` in example/experiment1.py:120-128)."""
        rows = range(self.shape[0])
        if isinstance(key, tuple) and len(key) == 2 and not (isinstance(key[1], slice) and key[1] == slice(None)):
            cols = range(self.shape[1])
            ci = (cols[key[1]],) if isinstance(key[1], int) else tuple(cols[key[1]])
            ri = (rows[key[0]],) if isinstance(key[0], int) else tuple(rows[key[0]])
            return Block(self, ri, ci)  # e[i, j], e[a:b, j], ...
        if isinstance(key, tuple) and len(key) == 2:
            key = key[0]  # e[i, :] / e[a:b, :] (``J(q)[0:2, :]``, example/planar_idk.py:42)
        if isinstance(key, int):
            idx = (rows[key],)
        elif isinstance(key, slice):
            idx = tuple(rows[key])
        else:
            raise NotImplementedError("only row indexing e[i] / e[a:b] (optionally with ', :') of a generic expression is supported")
        return Rows(self, idx)

    def numel(self) -> int:
        return self.shape[0] * self.shape[1]

    # linear / quadratic classification in the decision variables (cs.is_linear / is_quadratic,
    # builder.py:226-240,314-317,357-360)
    def degree(self) -> int:
        """0 constant/parameter, 1 linear in x, 2 quadratic in x, 3 'nonlinear'."""
        raise NotImplementedError


@dataclass(eq=False)
class Const(Expr):
    value: np.ndarray

    def __post_init__(self):
        self.value = np.atleast_2d(np.asarray(self.value, dtype=np.float64))
        if self.value.shape[0] == 1 and self.value.shape[1] > 1:
            self.value = self.value.T  # arrayify_args horzcat of a flat list gives a row; vectors are columns
        self.shape = self.value.shape

    def degree(self):
        return 0


def as_expr(x) -> Expr:
    if isinstance(x, Expr):
        return x
    return Const(np.asarray(x, dtype=np.float64))


@dataclass(eq=False)
class ParamRef(Expr):
    """A named parameter block (builder.add_parameter, builder.py:263-273)."""

    name: str
    m: int = 1
    n: int = 1

    def __post_init__(self):
        self.shape = (self.m, self.n)

    def degree(self):
        return 0

    def __getitem__(self, key):
        # obs[:, i]
        if isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None) and isinstance(key[1], int):
            return ParamCol(self, key[1] if key[1] >= 0 else self.n + key[1])
        return Expr.__getitem__(self, key)  # pg[3], pg[:3] (example/torque_control_example.py:72-75): rows of the block


@dataclass(eq=False)
class ParamCol(Expr):
    """One column of a parameter block."""

    param: "ParamRef" = None
    col: int = 0

    def __post_init__(self):
        self.shape = (self.param.m, 1)

    def degree(self):
        return 0


@dataclass(eq=False)
class StateRef(Expr):
    """Decision-variable block of a model: the whole trajectory (t is None) or one knot
    (builder.get_model_states / get_model_state, builder.py:124-149)."""

    var_name: str  # e.g. "kuka/q/x"
    model_name: str
    time_deriv: int
    m: int
    n: int
    t: Optional[int] = None

    def __post_init__(self):
        self.shape = (self.m, self.n if self.t is None else 1)

    def degree(self):
        return 1

    def __getitem__(self, key):
        # Q[:, t]
        if isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None) and isinstance(key[1], int) and self.t is None:
            t = key[1] if key[1] >= 0 else self.n + key[1]
            return StateRef(self.var_name, self.model_name, self.time_deriv, self.m, self.n, t)
        # Q[:, a:b]
        if isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None) and isinstance(key[1], slice) and self.t is None:
            lo, hi, step = key[1].indices(self.n)
            if step != 1:
                raise NotImplementedError("only unit-stride column slices are lowered")
            return StateCols(self, lo, hi)
        raise NotImplementedError("only Q[:, t] and Q[:, a:b] slicing is lowered")


@dataclass(eq=False)
class StateCols(Expr):
    """Columns lo..hi-1 of a state trajectory (dX[:, 1:], dX[:, :-1] in point_mass_mpc.py:134)."""

    state: "StateRef" = None
    lo: int = 0
    hi: int = 0

    def __post_init__(self):
        self.shape = (self.state.m, self.hi - self.lo)

    def degree(self):
        return 1


@dataclass(eq=False)
class RobotStates(Expr):
    """builder.get_robot_states_and_parameters (builder.py:178-204): the full dim x n joint trajectory of a robot assembled from its
    optimised block (decision variables) and its parameterised block (parameters), rows in actuated-joint order."""

    states: "StateRef" = None
    params: "ParamRef" = None
    opt_idx: tuple = ()
    par_idx: tuple = ()

    def __post_init__(self):
        self.shape = (len(self.opt_idx) + len(self.par_idx), self.states.shape[1])

    def degree(self):
        return 1


@dataclass(eq=False)
class Rows(Expr):
    """Row selection values[idx, :] (RobotModel.extract_optimized_dimensions / extract_parameter_dimensions, models.py:590-612)."""

    a: Expr = None
    idx: tuple = ()

    def __post_init__(self):
        self.shape = (len(self.idx), self.a.shape[1])

    def degree(self):
        return self.a.degree()


@dataclass(eq=False)
class Block(Expr):
    """Sub-matrix a[rows, cols]."""

    a: Expr = None
    ridx: tuple = ()
    cidx: tuple = ()

    def __post_init__(self):
        self.shape = (len(self.ridx), len(self.cidx))

    def degree(self):
        return self.a.degree()


@dataclass(eq=False)
class VarRef(Expr):
    """A free decision-variable block (builder.add_decision_variables, builder.py:244-261)."""

    var_name: str
    m: int = 1
    n: int = 1

    def __post_init__(self):
        self.shape = (self.m, self.n)

    def degree(self):
        return 1


@dataclass(eq=False)
class LinkFunction(Expr):
    """robot.get_global_link_{position,rotation,quaternion}(link, q) of a symbolic q
    (models.py:924-933, 986-995, 1049-1088; mapped over columns like .map(n), :786-787)."""

    robot: object
    link: str
    what: str  # "position" | "rotation" | "quaternion"
    q: Expr = None

    def __post_init__(self):
        rows = {"position": 3, "quaternion": 4, "rotation": 3, "geometric_jacobian": 6}[self.what]
        cols = self.q.shape[1]
        if self.what == "geometric_jacobian":
            if cols != 1:
                raise NotImplementedError("the Jacobian of a trajectory is a list in the reference; only one configuration is lowered")
            self.shape = (6, self.robot.ndof)
        elif self.what == "rotation":
            if cols != 1:
                raise NotImplementedError("rotation of a trajectory is a list in the reference; only one configuration is lowered")
            self.shape = (3, 3)
        else:
            self.shape = (rows, cols)

    def degree(self):
        return 0 if self.q.degree() == 0 else 3


@dataclass(eq=False)
class RneaFunction(Expr):
    """robot.rnea(q, qd, qdd) of symbolic trajectories (models.py:1731-1884; the reference's arrayify_args / SX graph is evaluated column by
    column): ndof x n inverse-dynamics torques."""

    robot: object
    q: Expr = None
    qd: Expr = None
    qdd: Expr = None

    def __post_init__(self):
        assert self.q.shape == self.qd.shape == self.qdd.shape, "rnea: q, qd, qdd must have the same shape"
        self.shape = self.q.shape

    def degree(self):
        return 0 if max(self.q.degree(), self.qd.degree(), self.qdd.degree()) == 0 else 3


@dataclass(eq=False)
class PathInFrame(Expr):
    """origin + R @ local[:, k] for every column k (figure_eight_plan.py:90-96)."""

    origin: Expr = None
    rotation: Expr = None
    local: np.ndarray = None

    def __post_init__(self):
        self.local = np.asarray(self.local, dtype=np.float64)
        assert self.local.shape[0] == 3
        self.shape = (3, self.local.shape[1])

    def degree(self):
        return max(self.origin.degree(), self.rotation.degree())


def _bshape(a: Expr, b: Expr):
    (ra, ca), (rb, cb) = a.shape, b.shape
    if ra != rb and 1 not in (ra * ca, rb * cb):
        raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")
    return (max(ra, rb), max(ca, cb))  # CasADi repeats a column / scalar (figure_eight_plan.py:107)


@dataclass(eq=False)
class Sub(Expr):
    a: Expr = None
    b: Expr = None

    def __post_init__(self):
        self.shape = _bshape(self.a, self.b)

    def degree(self):
        return max(self.a.degree(), self.b.degree())


@dataclass(eq=False)
class Add(Expr):
    a: Expr = None
    b: Expr = None

    def __post_init__(self):
        self.shape = _bshape(self.a, self.b)

    def degree(self):
        return max(self.a.degree(), self.b.degree())


@dataclass(eq=False)
class Scale(Expr):
    w: float = 1.0
    a: Expr = None

    def __post_init__(self):
        self.shape = self.a.shape

    def degree(self):
        return self.a.degree()


@dataclass(eq=False)
class SumSqr(Expr):
    a: Expr = None

    def __post_init__(self):
        self.shape = (1, 1)

    def degree(self):
        d = self.a.degree()
        return 0 if d == 0 else (2 if d == 1 else 3)


@dataclass(eq=False)
class MatMul(Expr):
    """Matrix product (``J @ qd``, example/experiment1.py:32)."""

    a: Expr = None
    b: Expr = None

    def __post_init__(self):
        assert self.a.shape[1] == self.b.shape[0], f"matmul shape mismatch {self.a.shape} @ {self.b.shape}"
        self.shape = (self.a.shape[0], self.b.shape[1])

    def degree(self):
        return min(3, self.a.degree() + self.b.degree())


@dataclass(eq=False)
class VCat(Expr):
    """casadi.vertcat of column blocks (``vertcat(desvel, zeros(4))``, example/experiment1.py:142)."""

    parts: tuple = ()

    def __post_init__(self):
        cols = {p.shape[1] for p in self.parts}
        assert len(cols) == 1, "vertcat needs equal column counts"
        self.shape = (sum(p.shape[0] for p in self.parts), cols.pop())

    def degree(self):
        return max(p.degree() for p in self.parts)


@dataclass(eq=False)
class Atan2(Expr):
    """casadi.atan2, elementwise (``2 * atan2(quat[2], quat[3])``: the planar heading, example/planar_idk.py:31)."""

    y: Expr = None
    x: Expr = None

    def __post_init__(self):
        self.shape = _bshape(self.y, self.x)

    def degree(self):
        return 0 if max(self.y.degree(), self.x.degree()) == 0 else 3


def atan2(y, x) -> Expr:
    return Atan2(as_expr(y), as_expr(x))


def vertcat(*parts) -> Expr:
    return VCat(tuple(as_expr(p) for p in parts))


@dataclass(eq=False)
class Mul(Expr):
    """Elementwise product with scalar broadcasting (``a * y`` in the reference's Booth test, tests/test_solver.py:31)."""

    a: Expr = None
    b: Expr = None

    def __post_init__(self):
        self.shape = _bshape(self.a, self.b)

    def degree(self):
        return min(3, self.a.degree() + self.b.degree())


@dataclass(eq=False)
class Square(Expr):
    """Elementwise square, ``(linkrad + obsrad) ** 2`` in builder.py:413."""

    a: Expr = None

    def __post_init__(self):
        self.shape = self.a.shape

    def degree(self):
        d = self.a.degree()
        return 0 if d == 0 else (2 if d == 1 else 3)


@dataclass(eq=False)
class Gather(Expr):
    """Signed re-arrangement of the entries of ``a``: out[r, c] = sign[r, c] * vec(a)[idx[r, c]] (column-major vec; idx < 0: a zero entry).
    One node for the structural operations of the reference's scripts -- transpose, ``spatialmath.skew`` of a symbolic vector
    (spatialmath.py:202-232), horizontal concatenation (as the transpose of a vertical one)."""

    a: Expr = None
    idx: np.ndarray = None
    sign: np.ndarray = None

    def __post_init__(self):
        self.idx = np.atleast_2d(np.asarray(self.idx, dtype=np.int64))
        self.sign = np.ones(self.idx.shape) if self.sign is None else np.atleast_2d(np.asarray(self.sign, dtype=np.float64))
        assert self.idx.shape == self.sign.shape and self.idx.max() < self.a.numel()
        self.sign = np.where(self.idx < 0, 0.0, self.sign)
        self.shape = self.idx.shape

    def degree(self):
        return self.a.degree()


def transpose(e) -> Expr:
    e = as_expr(e)
    m, n = e.shape
    if (m, n) == (1, 1):
        return e
    return Gather(e, np.arange(m * n).reshape(n, m))  # out[r, c] = a[c, r] = vec(a)[r * m + c]


def horzcat(*parts) -> Expr:
    """casadi.horzcat of blocks with equal row counts."""
    return transpose(VCat(tuple(transpose(q) for q in parts)))


def skew(v) -> Expr:
    """spatialmath.skew of a symbolic scalar or 3-vector (spatialmath.py:202-232): [[0, -z, y], [z, 0, -x], [-y, x, 0]]."""
    v = as_expr(v)
    if v.numel() == 1:
        return Gather(v, [[-1, 0], [0, -1]], [[0.0, -1.0], [1.0, 0.0]])
    if v.numel() != 3:
        raise ValueError("expecting a scalar or 3-vector")
    return Gather(v, [[-1, 2, 1], [2, -1, 0], [1, 0, -1]], [[0.0, -1.0, 1.0], [1.0, 0.0, -1.0], [-1.0, 1.0, 0.0]])


def _mtimes(a, b) -> Expr:
    """casadi.mtimes behind ``@``: a scalar operand multiplies elementwise (``Rc.T @ dt @ dp[:3]``, example/torque_control_example.py:69)."""
    if isinstance(a, (int, float)):
        return Scale(float(a), as_expr(b))
    if isinstance(b, (int, float)):
        return Scale(float(b), as_expr(a))
    a, b = as_expr(a), as_expr(b)
    if a.shape[1] != b.shape[0] and 1 in (a.numel(), b.numel()):
        return Mul(a, b)
    return MatMul(a, b)


def sumsqr(e) -> Expr:
    """casadi.sumsqr (re-exported by the reference, optas/__init__.py:2)."""
    return SumSqr(as_expr(e))


def path_in_frame(origin: Expr, rotation: Expr, local_path) -> Expr:
    """The loop ``path[:, k] = pc + Rc @ path[:, k]`` of figure_eight_plan.py:94-96 as one node."""
    return PathInFrame(as_expr(origin), as_expr(rotation), np.asarray(local_path, dtype=np.float64))
