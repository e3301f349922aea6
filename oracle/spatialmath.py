"""ORACLE (test infrastructure, not product code) -- numpy restatement of optas/spatialmath.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``optas_amd``) never does.

Every function cites the reference line it restates.  All arithmetic is float64 like CasADi's
``DM``/``SX``.  Pins: ``tests/test_oracle_spatialmath.py`` checks these against
``scipy.spatial.transform.Rotation`` exactly the way the reference's own
``tests/test_spatialmath.py`` does (that oracle *is* available in this image).
"""
import numpy as np

pi = np.pi  # spatialmath.py:15
eps = np.finfo(float).eps  # spatialmath.py:18


def I3():  # spatialmath.py:73-78
    return np.eye(3)


def I4():  # spatialmath.py:81-86
    return np.eye(4)


def unit(v):  # spatialmath.py:267-274  v / ||v||_F
    v = np.asarray(v, dtype=float).reshape(-1)
    return v / np.linalg.norm(v)


def skew(v):  # spatialmath.py:202-232
    v = np.asarray(v, dtype=float).reshape(-1)
    if v.shape[0] == 1:
        return np.array([[0.0, -v[0]], [v[0], 0.0]])
    if v.shape[0] == 3:
        return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    raise ValueError("expecting a scalar or 3-vector")


def angvec2r(theta, v):  # spatialmath.py:89-99  Rodrigues
    sk = skew(unit(v))
    return I3() + np.sin(theta) * sk + (1.0 - np.cos(theta)) * (sk @ sk)


def rotx(theta):  # spatialmath.py:115-127
    ct, st = np.cos(theta), np.sin(theta)
    return np.array([[1.0, 0.0, 0.0], [0.0, ct, -st], [0.0, st, ct]])


def roty(theta):  # spatialmath.py:130-142
    ct, st = np.cos(theta), np.sin(theta)
    return np.array([[ct, 0.0, st], [0.0, 1.0, 0.0], [-st, 0.0, ct]])


def rotz(theta):  # spatialmath.py:145-157
    ct, st = np.cos(theta), np.sin(theta)
    return np.array([[ct, -st, 0.0], [st, ct, 0.0], [0.0, 0.0, 1.0]])


def rpy2r(rpy, opt="zyx"):  # spatialmath.py:160-185
    r, p, y = np.asarray(rpy, dtype=float).reshape(-1)
    if opt in {"xyz", "arm"}:
        return rotx(y) @ roty(p) @ rotz(r)
    if opt in {"zyx", "vehicle"}:
        return rotz(y) @ roty(p) @ rotx(r)
    if opt in {"yxz", "camera"}:
        return roty(y) @ rotx(p) @ rotz(r)
    raise ValueError(f"didn't recognize given option {opt}")


def rt2tr(R, t):  # spatialmath.py:188-199
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t, dtype=float).reshape(-1)
    return T


def r2t(R):  # spatialmath.py:102-112
    T = np.eye(4)
    T[:3, :3] = R
    return T


def t2r(T):  # spatialmath.py:235-242
    return np.asarray(T)[:3, :3]


def transl(T):  # spatialmath.py:257-264
    return np.asarray(T)[:3, 3]


def invt(T):  # spatialmath.py:245-254
    R = t2r(T)
    t = transl(T)
    return rt2tr(R.T, -R.T @ t)


class Quaternion:
    """xyzw quaternion, spatialmath.py:277-404.  NB ``a * b`` is the *reversed* Hamilton product
    (rotation R(b)·R(a)), exactly as written at spatialmath.py:298-312."""

    def __init__(self, x, y, z, w):
        self._q = np.array([x, y, z, w], dtype=float)

    def split(self):
        return tuple(self._q)

    def __mul__(self, quat):  # spatialmath.py:298-312
        x0, y0, z0, w0 = self.split()
        x1, y1, z1, w1 = quat.split()
        return Quaternion(
            x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
            -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
            x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
            -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
        )

    def sumsqr(self):  # spatialmath.py:314-319
        return float(np.sum(self._q**2))

    def inv(self):  # spatialmath.py:321-328
        q = self._q
        qinv = np.concatenate([-q[:3], q[3:]]) / self.sumsqr()
        return Quaternion(*qinv)

    @staticmethod
    def fromrpy(rpy):  # spatialmath.py:330-349
        r, p, y = np.asarray(rpy, dtype=float).reshape(-1)
        cr, sr = np.cos(0.5 * r), np.sin(0.5 * r)
        cp, sp = np.cos(0.5 * p), np.sin(0.5 * p)
        cy, sy = np.cos(0.5 * y), np.sin(0.5 * y)
        x = sr * cp * cy - cr * sp * sy
        yy = cr * sp * cy + sr * cp * sy
        z = cr * cp * sy - sr * sp * cy
        w = cr * cp * cy + sr * sp * sy
        n = np.sqrt(x * x + yy * yy + z * z + w * w)
        return Quaternion(x / n, yy / n, z / n, w / n)

    @staticmethod
    def fromvec(q):  # spatialmath.py:351-362
        q = np.asarray(q, dtype=float).reshape(-1)
        return Quaternion(q[0], q[1], q[2], q[3])

    @staticmethod
    def fromangvec(theta, v):  # spatialmath.py:364-375
        w = np.cos(0.5 * theta)
        xyz = np.sin(0.5 * theta) * unit(v)
        return Quaternion(xyz[0], xyz[1], xyz[2], w)

    def getquat(self):  # spatialmath.py:377-382
        return self._q.copy()

    def getrpy(self):  # spatialmath.py:384-404 (keeps the sign-losing pi/2 branch)
        qx, qy, qz, qw = self.split()
        sinr_cosp = 2.0 * (qw * qx + qy * qz)
        cosr_cosp = 1.0 - 2.0 * (qx * qx + qy * qy)
        roll = np.arctan2(sinr_cosp, cosr_cosp)
        sinp = 2.0 * (qw * qy - qz * qx)
        pitch = pi / 2.0 if abs(sinp) >= 1.0 else np.arcsin(sinp)
        siny_cosp = 2.0 * (qw * qz + qx * qy)
        cosy_cosp = 1.0 - 2.0 * (qy * qy + qz * qz)
        yaw = np.arctan2(siny_cosp, cosy_cosp)
        return np.array([roll, pitch, yaw])
