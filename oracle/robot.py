"""ORACLE (test infrastructure, not product code) -- numpy restatement of the kinematics in
optas/models.py.  Literal, joint-by-joint, with the same loops the reference writes (including the
redundant FK-per-joint of the geometric Jacobian); no attempt at speed.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pins (see tests/test_oracle_models.py): the reference's tester robot closed form
Rz(q0)·Tx(2)·Rz(q1)·Tx(1)·Tz(q2)·Tz(.5) (tests/tester_robot.urdf, tests/test_models.py:206-250,
399-428, 461), quaternion FK vs scipy Rotation.from_matrix (the reference's own style of check,
tests/test_models.py:619-650), geometric Jacobian vs central differences.
"""
import json

import numpy as np

from .spatialmath import (
    I3,
    I4,
    Quaternion,
    angvec2r,
    invt,
    r2t,
    rpy2r,
    rt2tr,
    t2r,
    transl,
    unit,
)


class JointTypeNotSupported(NotImplementedError):  # models.py:217-230
    pass


class _J:
    __slots__ = ("name", "type", "parent", "child", "xyz", "rpy", "axis", "limit")


class OracleRobot:
    """Reads the ``*.kin.json`` constants (own loader: the oracle shares no code with optas_amd)."""

    def __init__(self, kin_json_filename=None, kin_dict=None, name=None):
        if kin_dict is None:
            with open(kin_json_filename, "r") as fh:
                kin_dict = json.load(fh)
        self.urdf_name = kin_dict["name"]
        self.name = name if name is not None else self.urdf_name  # models.py:318-319
        self.links = [l["name"] for l in kin_dict["links"]]
        self.link_inertials = {l["name"]: l.get("inertial") for l in kin_dict["links"]}
        self.joints = []
        for jd in kin_dict["joints"]:
            j = _J()
            j.name, j.type, j.parent, j.child = jd["name"], jd["type"], jd["parent"], jd["child"]
            j.xyz = jd.get("xyz")
            j.rpy = jd.get("rpy")
            j.axis = jd.get("axis")
            j.limit = jd.get("limit")
            self.joints.append(j)
        self.joint_map = {j.name: j for j in self.joints}

    # ---- urdf_parser_py surface used by the reference -----------------------------------------
    def get_root(self):
        children = {j.child for j in self.joints}
        roots = [l for l in self.links if l not in children]
        assert len(roots) == 1
        return roots[0]

    def get_chain(self, root, link):
        """Joint names root->link (urdf.get_chain(root, link, links=False), models.py:846)."""
        parent_joint = {j.child: j for j in self.joints}
        out = []
        cur = link
        while cur != root:
            j = parent_joint[cur]
            out.append(j.name)
            cur = j.parent
        return out[::-1]

    def add_base_frame(self, base_link, xyz=None, rpy=None, joint_name=None):  # models.py:552-588
        child_link = self.get_root()
        j = _J()
        j.name = joint_name if isinstance(joint_name, str) else base_link + "_and_" + child_link + "_joint"
        j.type, j.parent, j.child = "fixed", base_link, child_link
        j.xyz = [0.0] * 3 if xyz is None else list(xyz)
        j.rpy = [0.0] * 3 if rpy is None else list(rpy)
        j.axis, j.limit = None, None
        self.links.append(base_link)
        self.joints.append(j)
        self.joint_map[j.name] = j

    # ---- joint bookkeeping, models.py:332-550, 642-667 ----------------------------------------------
    @property
    def joint_names(self):
        return [j.name for j in self.joints]

    @property
    def link_names(self):
        return list(self.links)

    @property
    def actuated_joint_names(self):  # models.py:349-354 : document order of non-fixed joints
        return [j.name for j in self.joints if j.type != "fixed"]

    @property
    def ndof(self):  # models.py:414-420
        return len(self.actuated_joint_names)

    def get_actuated_joint_index(self, joint_name):  # models.py:661-667
        return self.actuated_joint_names.index(joint_name)

    def get_joint_origin(self, joint):  # models.py:642-651
        if joint.xyz is None:
            return np.zeros(3), np.zeros(3)
        return np.array(joint.xyz, dtype=float), np.array(joint.rpy, dtype=float)

    def get_joint_axis(self, joint):  # models.py:653-659
        axis = np.array(joint.axis, dtype=float) if joint.axis is not None else np.array([1.0, 0.0, 0.0])
        return unit(axis)

    @property
    def lower_actuated_joint_limits(self):  # models.py:438-446, 468-480
        return np.array([-1e9 if j.limit is None else j.limit["lower"] for j in self.joints if j.type != "fixed"])

    @property
    def upper_actuated_joint_limits(self):  # models.py:448-456, 482-494
        return np.array([1e9 if j.limit is None else j.limit["upper"] for j in self.joints if j.type != "fixed"])

    @property
    def velocity_actuated_joint_limits(self):  # models.py:458-466, 496-508
        return np.array([1e9 if j.limit is None else j.limit["velocity"] for j in self.joints if j.type != "fixed"])

    # ---- forward kinematics -------------------------------------------------------------------
    def get_global_link_transform(self, link, q):  # models.py:826-868
        q = np.asarray(q, dtype=float).reshape(-1)
        assert link in self.links, f"given link '{link}' does not appear in URDF"
        root = self.get_root()
        T = I4()
        if link == root:
            return T
        for joint_name in self.get_chain(root, link):
            joint = self.joint_map[joint_name]
            xyz, rpy = self.get_joint_origin(joint)
            if joint.type == "fixed":
                T = T @ rt2tr(rpy2r(rpy), xyz)
                continue
            qi = q[self.get_actuated_joint_index(joint.name)]
            T = T @ rt2tr(rpy2r(rpy), xyz)
            if joint.type in {"revolute", "continuous"}:
                T = T @ r2t(angvec2r(qi, self.get_joint_axis(joint)))
            elif joint.type == "prismatic":
                T = T @ rt2tr(I3(), qi * self.get_joint_axis(joint))
            else:
                raise JointTypeNotSupported(joint.type)
        return T

    def get_link_transform(self, link, q, base_link):  # models.py:884-898  (T_L · invt(T_B), sic)
        return self.get_global_link_transform(link, q) @ invt(self.get_global_link_transform(base_link, q))

    def get_global_link_position(self, link, q):  # models.py:924-933
        return transl(self.get_global_link_transform(link, q)).copy()

    def get_link_position(self, link, q, base_link):  # models.py:949-960
        return transl(self.get_link_transform(link, q, base_link)).copy()

    def get_global_link_rotation(self, link, q):  # models.py:986-995
        return t2r(self.get_global_link_transform(link, q)).copy()

    def get_link_rotation(self, link, q, base_link):  # models.py:1011-1023
        return t2r(self.get_link_transform(link, q, base_link)).copy()

    def get_global_link_quaternion(self, link, q):  # models.py:1049-1088
        q = np.asarray(q, dtype=float).reshape(-1)
        assert link in self.links
        root = self.get_root()
        quat = Quaternion(0.0, 0.0, 0.0, 1.0)
        if link == root:
            return quat.getquat()
        for joint_name in self.get_chain(root, link):
            joint = self.joint_map[joint_name]
            xyz, rpy = self.get_joint_origin(joint)
            if joint.type == "fixed":
                quat = Quaternion.fromrpy(rpy) * quat
                continue
            qi = q[self.get_actuated_joint_index(joint.name)]
            quat = Quaternion.fromrpy(rpy) * quat
            if joint.type in {"revolute", "continuous"}:
                quat = Quaternion.fromangvec(qi, self.get_joint_axis(joint)) * quat
            elif joint.type == "prismatic":
                pass
            else:
                raise JointTypeNotSupported(joint.type)
        return quat.getquat()

    def get_link_quaternion(self, link, q, base_link):  # models.py:1108-1122
        quat_L_W = Quaternion.fromvec(self.get_global_link_quaternion(link, q))
        quat_B_W = Quaternion.fromvec(self.get_global_link_quaternion(base_link, q))
        return (quat_L_W * quat_B_W.inv()).getquat()

    # ---- Jacobians ----------------------------------------------------------------------------
    def get_global_link_geometric_jacobian(self, link, q):  # models.py:1199-1264
        q = np.asarray(q, dtype=float).reshape(-1)
        root = self.get_root()
        e = self.get_global_link_position(link, q)
        chain = self.get_chain(root, link) if link != root else []
        joint_index_order = []
        jacobian_columns = []
        for joint in self.joints:
            if joint.type == "fixed":
                continue
            joint_index = self.get_actuated_joint_index(joint.name)
            joint_index_order.append(joint_index)
            qi = q[joint_index]
            if joint.name in chain:
                if joint.type in {"revolute", "continuous"}:
                    axis = self.get_joint_axis(joint)
                    R = self.get_global_link_rotation(joint.child, q)
                    R = R @ angvec2r(qi, axis)  # models.py:1233 (extra spin about its own axis)
                    p = self.get_global_link_position(joint.child, q)
                    z = R @ axis
                    pdot = np.cross(z, e - p)
                    jacobian_columns.append(np.concatenate([pdot, z]))
                elif joint.type == "prismatic":
                    axis = self.get_joint_axis(joint)
                    R = self.get_global_link_rotation(joint.child, q)
                    z = R @ axis
                    jacobian_columns.append(np.concatenate([z, np.zeros(3)]))
                else:
                    raise JointTypeNotSupported(joint.type)
            else:
                jacobian_columns.append(np.zeros(6))
        ordered = [jacobian_columns[idx] for idx in joint_index_order]  # models.py:1257
        return np.stack(ordered, axis=1)

    def get_link_geometric_jacobian(self, link, q, base_link):  # models.py:1320-1344
        J = self.get_global_link_geometric_jacobian(link, q)
        R = self.get_global_link_rotation(base_link, q).T
        K = np.zeros((6, 6))
        K[:3, :3] = R
        K[3:, 3:] = R
        return K @ J

    def get_global_link_linear_jacobian(self, link, q):  # models.py:1411-1423
        return self.get_global_link_geometric_jacobian(link, q)[:3, :]

    def get_global_link_angular_geometric_jacobian(self, link, q):  # models.py:1481-1494
        return self.get_global_link_geometric_jacobian(link, q)[3:, :]

    # ---- trajectory ("map(n)") helpers, models.py:729-824 -------------------------------------------
    def map_position(self, link, Q):
        Q = np.asarray(Q, dtype=float)
        return np.stack([self.get_global_link_position(link, Q[:, t]) for t in range(Q.shape[1])], axis=1)

    def map_quaternion(self, link, Q):
        Q = np.asarray(Q, dtype=float)
        return np.stack([self.get_global_link_quaternion(link, Q[:, t]) for t in range(Q.shape[1])], axis=1)

    def quaternion_batch(self, link, Q):
        """get_global_link_quaternion (models.py:1049-1088) for N configurations at once: Q (N, ndof) -> (N, 4) xyzw.  The same chain walk and
        the same reversed product (spatialmath.py:298-312) on component arrays; what the bench-scale tests check 13 M knots with
        (equal to the scalar restatement to rounding, tests/test_oracle_models.py)."""
        Q = np.atleast_2d(np.asarray(Q, dtype=float))
        N = Q.shape[0]

        def mul(a, b):  # Quaternion.__mul__: self = a, quat = b
            x0, y0, z0, w0 = a
            x1, y1, z1, w1 = b
            return (x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0, x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0, -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0)

        quat = (np.zeros(N), np.zeros(N), np.zeros(N), np.ones(N))
        root = self.get_root()
        if link != root:
            for joint_name in self.get_chain(root, link):
                joint = self.joint_map[joint_name]
                _, rpy = self.get_joint_origin(joint)
                quat = mul(tuple(Quaternion.fromrpy(rpy).getquat()), quat)
                if joint.type == "fixed" or joint.type == "prismatic":
                    continue
                if joint.type not in {"revolute", "continuous"}:
                    raise JointTypeNotSupported(joint.type)
                qi = Q[:, self.get_actuated_joint_index(joint.name)]
                ax = unit(self.get_joint_axis(joint))
                s = np.sin(0.5 * qi)
                quat = mul((s * ax[0], s * ax[1], s * ax[2], np.cos(0.5 * qi)), quat)
        return np.stack(quat, axis=1)

    def quaternion_jacobian(self, link, q):
        """d quat / d q (4x7), not a reference function: the reference gets it from CasADi AD of
        models.py:1049-1088.  For a Hamilton xyzw quaternion of R(q): dquat = 1/2 (omega,0) (x) quat with
        omega in the world frame, so column j is 1/2 (z_j, 0) (x) quat.  Checked against central
        differences of get_global_link_quaternion in tests/test_oracle_models.py."""
        quat = self.get_global_link_quaternion(link, q)
        Jw = self.get_global_link_angular_geometric_jacobian(link, q)
        x, y, z, w = quat
        out = np.zeros((4, Jw.shape[1]))
        for j in range(Jw.shape[1]):
            ox, oy, oz = Jw[:, j]
            # Hamilton product (ox,oy,oz,0) (x) (x,y,z,w), xyzw storage
            out[0, j] = 0.5 * (ox * w + oy * z - oz * y)
            out[1, j] = 0.5 * (-ox * z + oy * w + oz * x)
            out[2, j] = 0.5 * (ox * y - oy * x + oz * w)
            out[3, j] = 0.5 * (-ox * x - oy * y - oz * z)
        return out


# ---- inverse dynamics -------------------------------------------------------------------------------
def _skew3(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rnea_tables(robot: "OracleRobot"):
    """The data selection of RobotModel.rnea (models.py:1742-1784), quirks included: masses / centres of
    mass / inertias of the links that carry <inertial>, the FIRST one dropped (:1772-1774); joints of
    get_chain(root, link_names[-1]) with the FIRST one dropped (:1779-1782); inertial rpy ignored."""
    for j in robot.joints:  # models.py:1742-1746
        if j.type not in {"revolute", "continuous", "fixed"}:
            raise JointTypeNotSupported(j.type)
    if robot.joints[0].type != "fixed":  # models.py:1748-1749
        raise JointTypeNotSupported("First joint should be fixed")
    ine = [robot.link_inertials[l] for l in robot.links if robot.link_inertials[l] is not None]
    m = np.array([i["mass"] for i in ine][1:])
    cm = np.array([i["xyz"] for i in ine][1:]).T

    def mat(i):
        ixx, ixy, ixz, iyy, iyz, izz = i["inertia"]
        return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])

    Icm = [mat(i) for i in ine][1:]
    names = robot.get_chain(robot.get_root(), robot.links[-1])[1:]
    xyzs, rpys, axes = [], [], []
    for n in names:
        j = robot.joint_map[n]
        xyz, rpy = robot.get_joint_origin(j)
        xyzs.append(xyz)
        rpys.append(rpy)
        axes.append(robot.get_joint_axis(j))
    return m, cm, Icm, xyzs, rpys, axes


def rnea(robot: "OracleRobot", q, qd, qdd):
    """models.py:1731-1884, line by line (Craig ch. 6).  Returns tau for joints 0..n-2 of joints_list_r."""
    q, qd, qdd = (np.asarray(a, dtype=float).reshape(-1) for a in (q, qd, qdd))
    m, cm, Icm, xyzs, rpys, axes = rnea_tables(robot)
    n = len(xyzs)
    oms = [np.zeros(3)]
    omDs = [np.zeros(3)]
    vDs = [np.array([0.0, 0.0, 9.81])]  # -gravity_para, models.py:1789-1801
    fs = [np.zeros(3)]
    ns = [np.zeros(3)]
    for i in range(n):  # models.py:1819-1852
        if i != n - 1:
            iRp = (rpy2r(rpys[i]) @ angvec2r(q[i], axes[i])).T
            iaxisi = iRp @ axes[i]
            omi = iRp @ oms[i] + iaxisi * qd[i]
            omDi = iRp @ omDs[i] + _skew3(iRp @ oms[i]) @ (iaxisi * qd[i]) + iaxisi * qdd[i]
        else:
            iRp = rpy2r(rpys[i]).T
            omi = iRp @ oms[i]
            omDi = iRp @ omDs[i]
        vDi = iRp @ (vDs[i] + _skew3(omDs[i]) @ xyzs[i] + _skew3(oms[i]) @ (_skew3(oms[i]) @ xyzs[i]))
        fi = m[i] * (vDi + _skew3(omDi) @ cm[:, i] + _skew3(omi) @ (_skew3(omi) @ cm[:, i]))
        ni = Icm[i] @ omDi + _skew3(omi) @ Icm[i] @ omi
        oms.append(omi)
        omDs.append(omDi)
        vDs.append(vDi)
        fs.append(fi)
        ns.append(ni)
    ifi = fs[-1]  # models.py:1858-1859
    ini = ns[-1] + _skew3(cm[:, -1]) @ fs[-1]
    taus = []
    for i in range(n - 1, 0, -1):  # models.py:1863-1880
        if i < n - 1:
            pRi = rpy2r(rpys[i]) @ angvec2r(q[i], axes[i])
        else:
            pRi = rpy2r(rpys[i])
        ini = ns[i] + pRi @ ini + _skew3(cm[:, i - 1]) @ fs[i] + _skew3(xyzs[i]) @ pRi @ ifi
        ifi = pRi @ ifi + fs[i]
        pRi = rpy2r(rpys[i - 1]) @ angvec2r(q[i - 1], axes[i - 1])
        taus.append(float(ini @ pRi.T @ axes[i - 1]))
    return np.array(taus[::-1])
