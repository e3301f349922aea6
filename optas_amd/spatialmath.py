"""Host-side SE(3)/quaternion helpers (float64 numpy) with the reference's names and conventions
(optas/spatialmath.py).  They are used for *constants only* -- folding fixed URDF transforms into
the per-joint blocks that the HIP kernels consume -- and for user-side convenience.  Nothing
batched runs here: trajectories go through liboptas_hip.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

pi = math.pi
eps = float(np.finfo(float).eps)


def _v(x, n=None) -> np.ndarray:
    a = np.asarray(x, dtype=np.float64).reshape(-1)
    if n is not None and a.shape[0] != n:
        raise ValueError(f"expected {n} elements, got {a.shape[0]}")
    return a


def I3() -> np.ndarray:
    return np.eye(3)


def I4() -> np.ndarray:
    return np.eye(4)


def unit(v) -> np.ndarray:
    """v / ||v|| (spatialmath.py:267-274)."""
    a = _v(v)
    return a / math.sqrt(float(a @ a))


def _symbolic(*xs) -> bool:
    from .expr import Expr

    return any(isinstance(x, Expr) for x in xs)


def skew(v) -> np.ndarray:
    """Skew-symmetric matrix of a scalar or 3-vector (spatialmath.py:202-232); of an expression node when ``v`` is one."""
    if _symbolic(v):
        from .expr import skew as skew_node

        return skew_node(v)
    a = _v(v)
    if a.shape[0] == 1:
        return np.array([[0.0, -a[0]], [a[0], 0.0]])
    if a.shape[0] == 3:
        x, y, z = a
        return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    raise ValueError("expecting a scalar or 3-vector")


def angvec2r(theta: float, v) -> np.ndarray:
    """Rodrigues rotation (spatialmath.py:89-99)."""
    K = skew(unit(v))
    return np.eye(3) + math.sin(theta) * K + (1.0 - math.cos(theta)) * (K @ K)


def rotx(theta: float) -> np.ndarray:
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def roty(theta: float) -> np.ndarray:
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def rotz(theta: float) -> np.ndarray:
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rpy2r(rpy, opt: str = "zyx") -> np.ndarray:
    """Roll-pitch-yaw to SO(3); default "zyx" = Rz(y) Ry(p) Rx(r), the URDF convention
    (spatialmath.py:160-185)."""
    r, p, y = _v(rpy, 3)
    if opt in ("xyz", "arm"):
        return rotx(y) @ roty(p) @ rotz(r)
    if opt in ("zyx", "vehicle"):
        return rotz(y) @ roty(p) @ rotx(r)
    if opt in ("yxz", "camera"):
        return roty(y) @ rotx(p) @ rotz(r)
    raise ValueError(f"didn't recognize given option {opt}")


def rt2tr(R, t) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(R, dtype=np.float64)
    T[:3, 3] = _v(t, 3)
    return T


def r2t(R) -> np.ndarray:
    return rt2tr(R, np.zeros(3))


def t2r(T) -> np.ndarray:
    return np.asarray(T, dtype=np.float64)[:3, :3].copy()


def transl(T) -> np.ndarray:
    return np.asarray(T, dtype=np.float64)[:3, 3].copy()


def invt(T) -> np.ndarray:
    R, t = t2r(T), transl(T)
    return rt2tr(R.T, -R.T @ t)


class Quaternion:
    """xyzw quaternion.  ``a * b`` keeps the reference's operand order (spatialmath.py:298-312):
    the result represents the rotation R(b) R(a)."""

    __slots__ = ("_q", "_sym")

    # getrotm entry by entry as the reference defines it (spatialmath.py:426-437): constant + sum of coefficient * product of components
    # (0: x, 1: y, 2: z, 3: w).  Three entries differ from the textbook matrix of a unit quaternion -- [0][2] has x y for x z, [1][2]
    # w z for w x, [2][1] w w x for 2 w x -- and are kept, because the scripts' optima depend on what the reference computes, not on the
    # textbook (example/torque_control_example.py:72-77 passes its goal orientation through this function).
    _ROTM = (
        ((1.0, ((-2.0, (1, 1)), (-2.0, (2, 2)))), (0.0, ((2.0, (0, 1)), (-2.0, (3, 2)))), (0.0, ((2.0, (0, 1)), (2.0, (3, 1))))),
        ((0.0, ((2.0, (0, 1)), (2.0, (3, 2)))), (1.0, ((-2.0, (0, 0)), (-2.0, (2, 2)))), (0.0, ((2.0, (1, 2)), (-2.0, (3, 2))))),
        ((0.0, ((2.0, (0, 2)), (-2.0, (3, 1)))), (0.0, ((2.0, (1, 2)), (1.0, (3, 3, 0)))), (1.0, ((-2.0, (0, 0)), (-2.0, (1, 1))))),
    )

    def __init__(self, x: float, y: float, z: float, w: float):
        self._sym = (x, y, z, w) if _symbolic(x, y, z, w) else None  # components that are expression nodes (entries of a parameter)
        self._q = None if self._sym is not None else np.array([x, y, z, w], dtype=np.float64)

    def split(self):
        if self._sym is not None:
            return self._sym
        return tuple(float(v) for v in self._q)

    def getrotm(self):
        """3 x 3 matrix of the quaternion by the reference's formula (see _ROTM); an expression node when the components are."""
        comp = self.split()

        def entry(const, terms):
            acc = None
            for coef, factors in terms:
                prod = comp[factors[0]]
                for k in factors[1:]:
                    prod = prod * comp[k]
                term = coef * prod
                acc = term if acc is None else acc + term
            return acc + const if const else acc

        rows = [[entry(*e) for e in row] for row in self._ROTM]
        if self._sym is None:
            return np.array(rows, dtype=np.float64)
        from .expr import as_expr, horzcat, vertcat

        return vertcat(*[horzcat(*[as_expr(v) for v in row]) for row in rows])

    def _numbers(self) -> np.ndarray:
        if self._sym is not None:
            raise NotImplementedError("this Quaternion holds expression nodes: only split() and getrotm() are defined for it")
        return self._q

    def getquat(self) -> np.ndarray:
        return self._numbers().copy()

    def sumsqr(self) -> float:
        q = self._numbers()
        return float(q @ q)

    def __mul__(self, other: "Quaternion") -> "Quaternion":
        if not isinstance(other, Quaternion):
            raise AssertionError("unsupported type")
        x0, y0, z0, w0 = self._numbers()
        x1, y1, z1, w1 = other._numbers()
        return Quaternion(
            x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
            -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
            x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
            -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
        )

    def inv(self) -> "Quaternion":
        n2 = self.sumsqr()
        x, y, z, w = self._numbers()
        return Quaternion(-x / n2, -y / n2, -z / n2, w / n2)

    @staticmethod
    def fromvec(q: Sequence[float]) -> "Quaternion":
        a = _v(q, 4)
        return Quaternion(a[0], a[1], a[2], a[3])

    @staticmethod
    def fromrpy(rpy) -> "Quaternion":
        r, p, y = _v(rpy, 3)
        cr, sr = math.cos(0.5 * r), math.sin(0.5 * r)
        cp, sp = math.cos(0.5 * p), math.sin(0.5 * p)
        cy, sy = math.cos(0.5 * y), math.sin(0.5 * y)
        x = sr * cp * cy - cr * sp * sy
        yy = cr * sp * cy + sr * cp * sy
        z = cr * cp * sy - sr * sp * cy
        w = cr * cp * cy + sr * sp * sy
        n = math.sqrt(x * x + yy * yy + z * z + w * w)
        return Quaternion(x / n, yy / n, z / n, w / n)

    @staticmethod
    def fromangvec(theta: float, v) -> "Quaternion":
        a = unit(v) * math.sin(0.5 * theta)
        return Quaternion(a[0], a[1], a[2], math.cos(0.5 * theta))

    def getrpy(self) -> np.ndarray:
        x, y, z, w = self._numbers()
        roll = math.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
        sinp = 2.0 * (w * y - z * x)
        pitch = pi / 2.0 if abs(sinp) >= 1.0 else math.asin(sinp)
        yaw = math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
        return np.array([roll, pitch, yaw])
