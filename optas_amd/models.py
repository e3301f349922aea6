"""Host-side mirror of the reference's model classes (optas/models.py): ``Model``, ``TaskModel`` and
``RobotModel`` with the same names, argument meaning and error behaviour for everything the hot path
touches.  The reference builds CasADi SX graphs here; this mirror instead

* keeps the joint bookkeeping (names, indexes, limits: models.py:332-550, 642-667) on the host,
* folds the URDF chain into ``oh_chain`` constants (``kinematic_chain``), and
* evaluates forward kinematics / Jacobians numerically through liboptas_hip (``oh_fk_jac``) -- there
  is no CPU implementation behind these methods.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from . import _lib
from .expr import Expr, LinkFunction, Rows
from .spatialmath import Quaternion, rpy2r, unit
from .urdf import Joint, Link, RobotDescription, load_robot_description

ROBOTS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")


class JointTypeNotSupported(NotImplementedError):
    """Thrown for joint types other than fixed/revolute/continuous/prismatic (models.py:217-230)."""

    def __init__(self, joint_type: str):
        super().__init__(f"{joint_type} joints are currently not supported")


class Model:
    """models.py:79-186."""

    def __init__(self, name: str, dim: int, time_derivs: List[int], symbol: str, dlim: Dict[int, Tuple], T: Optional[int]):
        self.name = name
        self.dim = dim
        self.time_derivs = time_derivs
        self.symbol = symbol
        self.dlim = dlim
        self.T = T

    def get_name(self) -> str:
        return self.name

    def _check_deriv(self, time_deriv: int) -> None:
        assert (
            time_deriv in self.time_derivs
        ), f"Given time derivative time_deriv={time_deriv} is not recognized, only allowed {self.time_derivs}"

    def state_name(self, time_deriv: int) -> str:
        self._check_deriv(time_deriv)
        return self.name + "/" + "d" * time_deriv + self.symbol

    def state_parameter_name(self, time_deriv: int) -> str:
        self._check_deriv(time_deriv)
        return self.name + "/" + "d" * time_deriv + self.symbol + "/" + "p"

    def state_optimized_name(self, time_deriv: int) -> str:
        self._check_deriv(time_deriv)
        return self.name + "/" + "d" * time_deriv + self.symbol + "/" + "x"

    def get_limits(self, time_deriv: int):
        self._check_deriv(time_deriv)
        assert time_deriv in self.dlim.keys(), f"Limit for time derivative time_deriv={time_deriv} has not been given"
        return self.dlim[time_deriv]

    def in_limit(self, x, time_deriv: int) -> bool:
        lo, up = self.get_limits(time_deriv)
        x = np.asarray(x, dtype=np.float64).reshape(len(lo), -1)
        return bool(np.all((lo.reshape(-1, 1) <= x) & (x <= up.reshape(-1, 1))))


class TaskModel(Model):
    """models.py:189-214."""

    def __init__(self, name, dim, time_derivs=[0], symbol="y", dlim={}, T=None, is_discrete=False):
        super().__init__(name, dim, time_derivs, symbol, dlim, T)
        self.is_discrete = is_discrete


class RobotModel(Model):
    """models.py:233-321.  ``urdf_filename`` may be a URDF (XML) or a ``*.kin.json`` constants file;
    xacro is not processed (no ``xacro`` in this environment): pass the expanded URDF instead."""

    def __init__(
        self,
        urdf_filename: Optional[str] = None,
        urdf_string: Optional[str] = None,
        xacro_filename: Optional[str] = None,
        name: Optional[str] = None,
        time_derivs: List[int] = [0],
        qddlim=None,
        T: Optional[int] = None,
        param_joints: List[str] = [],
    ):
        if xacro_filename is not None:
            raise NotImplementedError("xacro processing is not available; expand the xacro to URDF first")
        self.urdf: Optional[RobotDescription] = None
        self.urdf_filename = None
        self.urdf_string = None
        if urdf_filename is not None:
            self.urdf_filename = urdf_filename
            self.urdf = load_robot_description(urdf_filename)
        if urdf_string is not None:
            self.urdf_string = urdf_string
            if urdf_string.lstrip().startswith("{"):  # the constants format of *.kin.json as a string
                import json as _json

                self.urdf = RobotDescription.from_dict(_json.loads(urdf_string))
            else:
                self.urdf = RobotDescription.from_xml_string(urdf_string)
        assert self.urdf is not None, "You need to supply a urdf, either through filename or as a string"
        self.param_joints = param_joints
        dlim = {
            0: (self.lower_optimized_joint_limits, self.upper_optimized_joint_limits),
            1: (-self.velocity_optimized_joint_limits, self.velocity_optimized_joint_limits),
        }
        if qddlim:
            qddlim = np.asarray(qddlim, dtype=np.float64).reshape(-1)
            if qddlim.shape[0] == 1:
                qddlim = qddlim * np.ones(self.ndof)
            assert qddlim.shape[0] == self.ndof, f"expected ddlim to have {self.ndof} elements"
            dlim[2] = (-qddlim, qddlim)
        if name is None:
            name = self.urdf.name
        super().__init__(name, self.ndof, time_derivs, "q", dlim, T)
        self._fk_handles: Dict[str, "KinematicsHandle"] = {}

    @staticmethod
    def from_description(description: RobotDescription, **kwargs) -> "RobotModel":
        """From an already parsed kinematic tree (e.g. converted from the reference's ``RobotModel.urdf``, optas_amd.probe_lowering)."""
        import json as _json

        return RobotModel(urdf_string=_json.dumps(description.to_dict()), **kwargs)

    @staticmethod
    def builtin(robot: str, **kwargs) -> "RobotModel":
        """Robots shipped as kinematic constants: ``kuka_lwr``, ``med7``."""
        return RobotModel(urdf_filename=os.path.join(ROBOTS_DIR, robot + ".kin.json"), **kwargs)

    def get_urdf(self) -> RobotDescription:
        return self.urdf

    # ---- names / indexes (models.py:332-436) ------------------------------------------------------
    @property
    def joint_names(self) -> List[str]:
        return [j.name for j in self.urdf.joints]

    @property
    def link_names(self) -> List[str]:
        return [l.name for l in self.urdf.links]

    @property
    def actuated_joint_names(self) -> List[str]:
        return [j.name for j in self.urdf.joints if j.type != "fixed"]

    @property
    def parameter_joint_names(self) -> List[str]:
        return [j for j in self.actuated_joint_names if j in self.param_joints]

    @property
    def optimized_joint_names(self) -> List[str]:
        return [j for j in self.actuated_joint_names if j not in self.parameter_joint_names]

    @property
    def optimized_joint_indexes(self) -> List[int]:
        return [self.get_actuated_joint_index(j) for j in self.optimized_joint_names]

    @property
    def parameter_joint_indexes(self) -> List[int]:
        return [self.get_actuated_joint_index(j) for j in self.parameter_joint_names]

    @property
    def ndof(self) -> int:
        return len(self.actuated_joint_names)

    @property
    def num_opt_joints(self) -> int:
        return len(self.optimized_joint_names)

    @property
    def num_param_joints(self) -> int:
        return len(self.parameter_joint_names)

    def get_actuated_joint_index(self, joint_name: str) -> int:
        return self.actuated_joint_names.index(joint_name)

    # ---- limits (models.py:438-550) ---------------------------------------------------------------
    @staticmethod
    def get_joint_lower_limit(joint: Joint) -> float:
        return -1e9 if joint.limit is None else joint.limit.lower

    @staticmethod
    def get_joint_upper_limit(joint: Joint) -> float:
        return 1e9 if joint.limit is None else joint.limit.upper

    @staticmethod
    def get_velocity_joint_limit(joint: Joint) -> float:
        return 1e9 if joint.limit is None else joint.limit.velocity

    def _limits(self, getter, names) -> np.ndarray:
        return np.array([getter(j) for j in self.urdf.joints if j.name in names], dtype=np.float64)

    @property
    def lower_actuated_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_joint_lower_limit, self.actuated_joint_names)

    @property
    def upper_actuated_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_joint_upper_limit, self.actuated_joint_names)

    @property
    def velocity_actuated_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_velocity_joint_limit, self.actuated_joint_names)

    @property
    def lower_optimized_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_joint_lower_limit, self.optimized_joint_names)

    @property
    def upper_optimized_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_joint_upper_limit, self.optimized_joint_names)

    @property
    def velocity_optimized_joint_limits(self) -> np.ndarray:
        return self._limits(self.get_velocity_joint_limit, self.optimized_joint_names)

    def extract_parameter_dimensions(self, values):
        if isinstance(values, Expr):
            return Rows(values, tuple(self.parameter_joint_indexes))
        a = np.asarray(values)
        return a[self.parameter_joint_indexes] if a.ndim == 1 else a[self.parameter_joint_indexes, :]

    def extract_optimized_dimensions(self, values):
        if isinstance(values, Expr):
            return Rows(values, tuple(self.optimized_joint_indexes))
        a = np.asarray(values)
        return a[self.optimized_joint_indexes] if a.ndim == 1 else a[self.optimized_joint_indexes, :]

    # ---- tree (models.py:552-667) -----------------------------------------------------------------
    def add_base_frame(self, base_link: str, xyz=None, rpy=None, joint_name: Optional[str] = None) -> None:
        child_link = self.urdf.get_root()
        xyz = [0.0] * 3 if xyz is None else [float(v) for v in xyz]
        rpy = [0.0] * 3 if rpy is None else [float(v) for v in rpy]
        if not isinstance(joint_name, str):
            joint_name = base_link + "_and_" + child_link + "_joint"
        self.urdf.add_link(Link(name=base_link))
        self.urdf.add_joint(Joint(name=joint_name, type="fixed", parent=base_link, child=child_link, xyz=xyz, rpy=rpy))
        self._fk_handles.clear()

    def get_root_link(self) -> str:
        return self.urdf.get_root()

    @staticmethod
    def get_joint_origin(joint: Joint) -> Tuple[np.ndarray, np.ndarray]:
        if joint.xyz is None:
            return np.zeros(3), np.zeros(3)
        return np.array(joint.xyz, dtype=np.float64), np.array(joint.rpy, dtype=np.float64)

    @staticmethod
    def get_joint_axis(joint: Joint) -> np.ndarray:
        return unit(joint.axis if joint.axis is not None else [1.0, 0.0, 0.0])

    # ---- lowering: URDF chain -> oh_chain ---------------------------------------------------------
    def link_attachments(self, link: str, link_names) -> list:
        """For each named link on the chain root->link: (index of the last actuated chain joint before it, position of the
        link origin in the frame that follows that joint's motion) -- what oh_guards.link_joint / link_offset carry
        (sphere centres of sphere_collision_avoidance_constraints, builder.py:366-417)."""
        root = self.urdf.get_root()
        att = {root: (-1, np.zeros(3))}
        R_acc, p_acc = np.eye(3), np.zeros(3)
        k = 0
        names = self.urdf.get_chain(root, link, links=False) if link != root else []
        for name in names:
            joint = self.urdf.joint_map[name]
            xyz, rpy = self.get_joint_origin(joint)
            p_acc = p_acc + R_acc @ xyz
            R_acc = R_acc @ rpy2r(rpy)
            if joint.type == "fixed":
                att[joint.child] = (k - 1, p_acc.copy())
                continue
            att[joint.child] = (k, np.zeros(3))
            R_acc, p_acc = np.eye(3), np.zeros(3)
            k += 1
        out = []
        for ln in link_names:
            if ln not in att:
                raise ValueError(f"link '{ln}' is not on the chain from '{root}' to '{link}'")
            out.append(att[ln])
        return out

    def solver_chain(self, link: str) -> _lib.oh_chain:
        """The chain the solver kernels walk: kinematic_chain(link) for a model without parameterised joints; with
        param_joints = [the first actuated joint of the chain] the chain lists the optimised joints only (ndof = num_opt_joints,
        q index = optimised index) and the parameterised joint becomes the lead joint (oh_chain.has_lead)."""
        if self.num_param_joints == 0:
            return self.kinematic_chain(link)
        if self.num_param_joints != 1:
            raise NotImplementedError("only one parameterised joint is lowered")
        full = self.kinematic_chain(link)
        lead_i = self.parameter_joint_indexes[0]
        if full.n_chain < 2 or full.qidx[0] != lead_i or full.jtype[0] != 0:
            raise NotImplementedError("the parameterised joint must be the first (revolute) actuated joint of the chain")
        ch = _lib.oh_chain()
        ch.ndof = self.num_opt_joints
        ch.n_chain = full.n_chain - 1
        opt = self.optimized_joint_indexes
        for k in range(1, full.n_chain):
            ch.jtype[k - 1], ch.axcode[k - 1], ch.r0ident[k - 1] = full.jtype[k], full.axcode[k], full.r0ident[k]
            ch.qidx[k - 1] = opt.index(full.qidx[k])
            for i in range(9):
                ch.R0[k - 1][i] = full.R0[k][i]
            for i in range(3):
                ch.p0[k - 1][i], ch.axis[k - 1][i] = full.p0[k][i], full.axis[k][i]
            for i in range(4):
                ch.quat0[k - 1][i] = full.quat0[k][i]
        for i in range(9):
            ch.R_tool[i], ch.lead_R0[i] = full.R_tool[i], full.R0[0][i]
        for i in range(3):
            ch.p_tool[i], ch.lead_p0[i], ch.lead_axis[i] = full.p_tool[i], full.p0[0][i], full.axis[0][i]
        for i in range(4):
            ch.quat_tool[i] = full.quat_tool[i]
        ch.has_lead, ch.lead_axcode = 1, full.axcode[0]
        return ch

    def kinematic_chain(self, link: str) -> _lib.oh_chain:
        """Fold root->link into per-actuated-joint constants (fixed joints multiplied into the next
        actuated joint's pre-transform; trailing fixed joints into the tool transform), in the order
        get_global_link_transform (models.py:846-866) walks them."""
        assert link in self.urdf.link_map.keys(), f"given link '{link}' does not appear in URDF"
        root = self.urdf.get_root()
        ch = _lib.oh_chain()
        ch.ndof = self.ndof
        R_acc, p_acc = np.eye(3), np.zeros(3)
        quat_acc = Quaternion(0.0, 0.0, 0.0, 1.0)
        k = 0
        jmap = self.urdf.joint_map
        names = self.urdf.get_chain(root, link, links=False) if link != root else []
        for name in names:
            joint = jmap[name]
            xyz, rpy = self.get_joint_origin(joint)
            p_acc = p_acc + R_acc @ xyz
            R_acc = R_acc @ rpy2r(rpy)
            quat_acc = Quaternion.fromrpy(rpy) * quat_acc
            if joint.type == "fixed":
                continue
            if joint.type in ("revolute", "continuous"):
                jt = 0
            elif joint.type == "prismatic":
                jt = 1
            else:
                raise JointTypeNotSupported(joint.type)
            if k >= _lib.OH_MAX_CHAIN:
                raise ValueError(f"chain to '{link}' has more than {_lib.OH_MAX_CHAIN} actuated joints")
            ch.jtype[k] = jt
            ch.qidx[k] = self.get_actuated_joint_index(joint.name)
            axis = self.get_joint_axis(joint)
            code = 0
            for m in range(3):
                e = np.zeros(3)
                e[m] = 1.0
                if np.array_equal(axis, e):
                    code = m + 1
                elif np.array_equal(axis, -e):
                    code = -(m + 1)
            ch.axcode[k] = code
            ch.r0ident[k] = 1 if np.array_equal(R_acc, np.eye(3)) else 0
            for i in range(9):
                ch.R0[k][i] = float(R_acc.reshape(-1)[i])
            for i in range(3):
                ch.p0[k][i] = float(p_acc[i])
                ch.axis[k][i] = float(axis[i])
            qv = quat_acc.getquat()
            for i in range(4):
                ch.quat0[k][i] = float(qv[i])
            R_acc, p_acc = np.eye(3), np.zeros(3)
            quat_acc = Quaternion(0.0, 0.0, 0.0, 1.0)
            k += 1
        ch.n_chain = k
        for i in range(9):
            ch.R_tool[i] = float(R_acc.reshape(-1)[i])
        for i in range(3):
            ch.p_tool[i] = float(p_acc[i])
        qv = quat_acc.getquat()
        for i in range(4):
            ch.quat_tool[i] = float(qv[i])
        return ch

    # ---- inverse dynamics (models.py:1731-1884) --------------------------------------------------------------
    def dynamics_tables(self) -> _lib.oh_dynamics:
        """The constants RobotModel.rnea gathers (models.py:1742-1784), with the reference's selection
        rules and error behaviour: only revolute/continuous/fixed joints, first URDF joint fixed, masses of
        the links that carry <inertial> with the first dropped, chain to the last link with the first joint
        dropped, inertial rpy ignored."""
        for joint in self.urdf.joints:
            if joint.type not in {"revolute", "continuous", "fixed"}:
                raise JointTypeNotSupported(joint.type)
        if self.urdf.joints[0].type != "fixed":
            raise JointTypeNotSupported("First joint should be fixed")
        ine = [l.inertial for l in self.urdf.links if l.inertial is not None][1:]
        names = self.urdf.get_chain(self.urdf.get_root(), self.link_names[-1], links=False)[1:]
        n = len(names)
        if n < 2 or n > _lib.OH_MAX_BODIES - 1 or len(ine) < n:
            raise ValueError(f"rnea: unsupported chain ({n} bodies, {len(ine)} inertial links)")
        d = _lib.oh_dynamics()
        d.n, d.ndof = n, n - 1
        jm = self.urdf.joint_map
        for i, name in enumerate(names):
            xyz, rpy = self.get_joint_origin(jm[name])
            R = rpy2r(rpy).reshape(-1)
            axis = self.get_joint_axis(jm[name])
            ixx, ixy, ixz, iyy, iyz, izz = ine[i].inertia
            I = [ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz]
            for k in range(9):
                d.R0[i][k] = float(R[k])
                d.inertia[i][k] = float(I[k])
            for k in range(3):
                d.xyz[i][k] = float(xyz[k])
                d.axis[i][k] = float(axis[k])
                d.com[i][k] = float(ine[i].xyz[k])
            d.mass[i] = float(ine[i].mass)
        d.vd0[0], d.vd0[1], d.vd0[2] = 0.0, 0.0, 9.81
        return d

    def rnea(self, q, qd, qdd) -> np.ndarray:
        """Inverse dynamics tau(q, qd, qdd) (models.py:1731-1884), evaluated by liboptas_hip (oh_rnea).
        Arguments are ndof vectors or ndof-by-n arrays (columns = samples); symbolic arguments give the expression node the
        torque-MPC family is lowered from (h = TAU - rnea(Q, dQ, ddQ))."""
        import ctypes as C

        from .expr import Expr, RneaFunction, as_expr

        if any(isinstance(a, Expr) for a in (q, qd, qdd)):
            self.dynamics_tables()  # the reference's precondition checks (models.py:1742-1749) fire at graph-construction time
            return RneaFunction(self, as_expr(q), as_expr(qd), as_expr(qdd))

        if getattr(self, "_dyn_handle", None) is None:
            lib = _lib.load()
            dyn = self.dynamics_tables()
            desc = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_KINEMATICS, ndof=max(1, min(dyn.ndof, _lib.OH_MAX_CHAIN)))
            h = C.c_void_p()
            _lib.check(lib.oh_create(C.byref(desc), C.byref(h)), "oh_create")
            _lib.check(lib.oh_set_dynamics(h, C.byref(dyn)), "oh_set_dynamics")
            self._dyn_handle, self._dyn = h, dyn
        nd = self._dyn.ndof
        single = np.asarray(q).ndim == 1
        A = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(nd, -1).T) for a in (q, qd, qdd)]
        n = A[0].shape[0]
        tau = np.empty((n, nd))
        _lib.check(_lib.load().oh_rnea(self._dyn_handle, n, _lib._ptr(A[0]), _lib._ptr(A[1]), _lib._ptr(A[2]), _lib._ptr(tau)), "oh_rnea")
        return tau[0] if single else tau.T

    def rnea_jacobian(self, q, qd, qdd) -> np.ndarray:
        """d tau / d (q, qd, qdd) at ndof-by-n sample columns: (n, ndof, 3 ndof), exact (oh_rnea_jac)."""
        import ctypes as C

        self.rnea(np.zeros(self.ndof), np.zeros(self.ndof), np.zeros(self.ndof))  # creates the handle
        nd = self._dyn.ndof
        A = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(nd, -1).T) for a in (q, qd, qdd)]
        n = A[0].shape[0]
        J = np.empty((n, nd, 3 * nd))
        _lib.check(_lib.load().oh_rnea_jac(self._dyn_handle, n, _lib._ptr(A[0]), _lib._ptr(A[1]), _lib._ptr(A[2]), _lib._ptr(J)), "oh_rnea_jac")
        return J

    def rnea_hessian(self, q, qd, qdd, c) -> np.ndarray:
        """sum_i c_i d^2 tau_i / d (q, qd, qdd)^2 at ndof-by-n sample columns (c: ndof-by-n multipliers): (n, 3 ndof, 3 ndof), exact (oh_rnea_hess);
        what the reference's AD gives as the dynamics rows' share of ddh / the Lagrangian Hessian (optimization.py:8-24)."""
        self.rnea(np.zeros(self.ndof), np.zeros(self.ndof), np.zeros(self.ndof))  # creates the handle
        nd = self._dyn.ndof
        A = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(nd, -1).T) for a in (q, qd, qdd, c)]
        n = A[0].shape[0]
        H = np.empty((n, 3 * nd, 3 * nd))
        _lib.check(_lib.load().oh_rnea_hess(self._dyn_handle, n, _lib._ptr(A[0]), _lib._ptr(A[1]), _lib._ptr(A[2]), _lib._ptr(A[3]), _lib._ptr(H)), "oh_rnea_hess")
        return H

    # ---- numeric kinematics through the HIP library -------------------------------------------------
    def _kin(self, link: str) -> "KinematicsHandle":
        h = self._fk_handles.get(link)
        if h is None:
            h = KinematicsHandle(self.kinematic_chain(link))
            self._fk_handles[link] = h
        return h

    def _q_cols(self, q) -> np.ndarray:
        a = np.asarray(q, dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        assert a.shape[0] == self.ndof, f"expected {self.ndof} rows, got {a.shape[0]}"
        return a

    def get_global_link_position(self, link: str, q) -> np.ndarray:
        """models.py:924-933; q is ndof or ndof-by-n (columns = joint states, like the reference).
        A symbolic q (builder state / parameter) returns an expression node instead."""
        if isinstance(q, Expr):
            return LinkFunction(self, link, "position", q)
        Q = self._q_cols(q)
        pose, _ = self._kin(link).fk_jac(Q.T, want_jac=False)
        out = pose[:, :3].T
        return out[:, 0] if np.asarray(q).ndim == 1 else out

    def get_global_link_quaternion(self, link: str, q) -> np.ndarray:
        """models.py:1049-1088 (xyzw, the reference's sign)."""
        if isinstance(q, Expr):
            return LinkFunction(self, link, "quaternion", q)
        Q = self._q_cols(q)
        pose, _ = self._kin(link).fk_jac(Q.T, want_jac=False)
        out = pose[:, 3:].T
        return out[:, 0] if np.asarray(q).ndim == 1 else out

    def get_global_link_rotation(self, link: str, q):
        """models.py:986-995.  Numerically the rotation is rebuilt from the (reference-signed)
        quaternion that oh_fk_jac returns."""
        if isinstance(q, Expr):
            return LinkFunction(self, link, "rotation", q)
        quat = self.get_global_link_quaternion(link, q)

        def q2r(v):
            x, y, z, w = v
            return np.array(
                [
                    [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                    [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                    [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
                ]
            )

        return q2r(quat) if quat.ndim == 1 else [q2r(quat[:, i]) for i in range(quat.shape[1])]

    def get_global_link_geometric_jacobian(self, link: str, q):
        """models.py:1199-1264: 6 x ndof (a list of them for a trajectory)."""
        if isinstance(q, Expr):
            return LinkFunction(self, link, "geometric_jacobian", q)
        Q = self._q_cols(q)
        _, J = self._kin(link).fk_jac(Q.T, want_pose=False)
        return J[0] if np.asarray(q).ndim == 1 else [J[i] for i in range(J.shape[0])]

    def get_global_link_linear_jacobian(self, link: str, q):
        if isinstance(q, Expr):
            return LinkFunction(self, link, "geometric_jacobian", q)[0:3]
        J = self.get_global_link_geometric_jacobian(link, q)
        return J[:3] if isinstance(J, np.ndarray) else [j[:3] for j in J]

    def get_global_link_angular_geometric_jacobian(self, link: str, q):
        if isinstance(q, Expr):
            return LinkFunction(self, link, "geometric_jacobian", q)[3:6]
        J = self.get_global_link_geometric_jacobian(link, q)
        return J[3:] if isinstance(J, np.ndarray) else [j[3:] for j in J]

    # ---- 4x4 transforms and base-frame variants (models.py:826-868, 884-898, 949-960, 1011-1023, 1108-1122, 1320-1344, 1425-1443,
    # 1496-1516): host compositions of the oh_fk_jac outputs, one configuration at a time ---------------------------------
    def get_global_link_transform(self, link: str, q) -> np.ndarray:
        """models.py:826-868: homogeneous transform of the link in the root frame."""
        T = np.eye(4)
        T[:3, :3] = self.get_global_link_rotation(link, np.asarray(q, dtype=np.float64).reshape(-1))
        T[:3, 3] = self.get_global_link_position(link, np.asarray(q, dtype=np.float64).reshape(-1))
        return T

    def get_link_transform(self, link: str, q, base_link: str) -> np.ndarray:
        """models.py:884-898: T_L invt(T_B) -- the reference's convention (not invt(T_B) T_L), pinned by its tests."""
        TB = self.get_global_link_transform(base_link, q)
        inv = np.eye(4)
        inv[:3, :3] = TB[:3, :3].T
        inv[:3, 3] = -TB[:3, :3].T @ TB[:3, 3]
        return self.get_global_link_transform(link, q) @ inv

    def get_link_position(self, link: str, q, base_link: str) -> np.ndarray:
        return self.get_link_transform(link, q, base_link)[:3, 3].copy()

    def get_link_rotation(self, link: str, q, base_link: str) -> np.ndarray:
        return self.get_link_transform(link, q, base_link)[:3, :3].copy()

    def get_link_quaternion(self, link: str, q, base_link: str) -> np.ndarray:
        """models.py:1108-1122: quat_L * quat_B^{-1} with the reference's reversed product (spatialmath.py:298-312)."""
        qv = np.asarray(q, dtype=np.float64).reshape(-1)
        ql = Quaternion.fromvec(self.get_global_link_quaternion(link, qv))
        qb = Quaternion.fromvec(self.get_global_link_quaternion(base_link, qv))
        return (ql * qb.inv()).getquat()

    def get_link_geometric_jacobian(self, link: str, q, base_link: str) -> np.ndarray:
        """models.py:1320-1344: blkdiag(R_B^T, R_B^T) J."""
        qv = np.asarray(q, dtype=np.float64).reshape(-1)
        J = self.get_global_link_geometric_jacobian(link, qv)
        RT = self.get_global_link_rotation(base_link, qv).T
        return np.vstack([RT @ J[:3], RT @ J[3:]])

    def get_link_linear_jacobian(self, link: str, q, base_link: str) -> np.ndarray:
        return self.get_link_geometric_jacobian(link, q, base_link)[:3]

    def get_link_angular_geometric_jacobian(self, link: str, q, base_link: str) -> np.ndarray:
        return self.get_link_geometric_jacobian(link, q, base_link)[3:]

    def get_link_position_function(self, link: str, base_link: str, n: int = 1, numpy_output: bool = True):
        """models.py:962-984 (what sphere_collision_avoidance_constraints maps over the knots)."""
        return lambda q: self.get_link_position(link, q, base_link)

    def get_global_link_transform_function(self, link: str, n: int = 1, numpy_output: bool = True):
        return lambda q: self.get_global_link_transform(link, q)

    def get_global_link_position_function(self, link: str, n: int = 1, numpy_output: bool = True):
        """models.py:935-947: callable on an ndof-by-n array -> 3-by-n."""
        return lambda Q: self.get_global_link_position(link, Q if isinstance(Q, Expr) else np.asarray(Q, dtype=np.float64).reshape(self.ndof, -1))

    def get_global_link_quaternion_function(self, link: str, n: int = 1, numpy_output: bool = True):
        return lambda Q: self.get_global_link_quaternion(link, Q if isinstance(Q, Expr) else np.asarray(Q, dtype=np.float64).reshape(self.ndof, -1))

    def get_global_link_linear_jacobian_function(self, link: str, n: int = 1, numpy_output: bool = True):
        return lambda Q: self.get_global_link_linear_jacobian(link, Q if isinstance(Q, Expr) else np.asarray(Q, dtype=np.float64).reshape(-1))

    def get_global_link_angular_geometric_jacobian_function(self, link: str, n: int = 1, numpy_output: bool = True):
        return lambda Q: self.get_global_link_angular_geometric_jacobian(link, Q if isinstance(Q, Expr) else np.asarray(Q, dtype=np.float64).reshape(-1))

    def get_global_link_geometric_jacobian_function(self, link: str, n: int = 1, numpy_output: bool = True):
        return lambda Q: self.get_global_link_geometric_jacobian(link, np.asarray(Q, dtype=np.float64).reshape(self.ndof, -1))


class KinematicsHandle:
    """A liboptas_hip handle used only for oh_fk_jac on one chain."""

    def __init__(self, chain: _lib.oh_chain):
        import ctypes as C

        lib = _lib.load()
        self.chain = chain
        self._h = None
        if chain.n_chain == 0:
            return  # the root link, or a link rigidly attached to it: a constant transform, nothing to launch
        desc = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_KINEMATICS, ndof=chain.ndof)
        self._h = C.c_void_p()
        _lib.check(lib.oh_create(C.byref(desc), C.byref(self._h)), "oh_create")
        _lib.check(lib.oh_set_constants(self._h, C.byref(chain)), "oh_set_constants")

    def fk_jac(self, Q: np.ndarray, want_pose: bool = True, want_jac: bool = True):
        """Q: n-by-ndof (row = one joint state, the ABI layout)."""
        lib = _lib.load()
        Q = _lib.as_f64(Q)
        n, ndof = Q.shape
        pose = np.empty((n, 7)) if want_pose else None
        J = np.empty((n, 6, ndof)) if want_jac else None
        if self._h is None:
            if want_pose:
                pose[:, :3] = np.array(self.chain.p_tool[:])
                pose[:, 3:] = np.array(self.chain.quat_tool[:])
            if want_jac:
                J[:] = 0.0
            return pose, J
        _lib.check(lib.oh_fk_jac(self._h, n, _lib._ptr(Q), _lib._ptr(pose), _lib._ptr(J)), "oh_fk_jac")
        return pose, J

    def __del__(self):
        try:
            if self._h:
                _lib.load().oh_destroy(self._h)
                self._h = None
        except Exception:
            pass
