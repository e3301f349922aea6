"""ctypes binding of liboptas_hip.so (include/optas_hip.h).  No torch, no numpy-side compute.

The library is built in-tree by ``__graft_entry__.build()`` / ``optas_amd.build``.  There is no CPU
fallback: if the shared object is missing, or no HIP device is present, every compute entry point
raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

OH_MAX_CHAIN = 16
OH_MAX_T = 256
OH_MAX_SPHERE_LINKS = 8
OH_MAX_OBSTACLES = 16
OH_COMM_ID_BYTES = 128

OH_OK, OH_ERR_INVALID, OH_ERR_HIP, OH_ERR_STATE = 0, 1, 2, 3
OH_ABI_VERSION = 7  # include/optas_hip.h: the struct layouts below are those of this version
OH_STATUS_CONVERGED, OH_STATUS_MAX_ITER, OH_STATUS_NUMERICAL, OH_STATUS_INFEASIBLE, OH_STATUS_ACCEPTABLE = 0, 1, 2, 3, 4
# IPOPT's names for the same outcomes (what CasADiSolver.stats()["return_status"] holds, solver.py:407-412)
STATUS_NAMES = {0: "Solve_Succeeded", 1: "Maximum_Iterations_Exceeded", 2: "Numerical_Failure", 3: "Infeasible_Problem_Detected", 4: "Solved_To_Acceptable_Level"}
_SEVERITY = np.array([0, 2, 3, 4, 1])  # by status code: converged < acceptable < iteration cap < numerical < infeasible


def status_ok(status) -> np.ndarray:
    """did_solve() per instance: converged, or solved to the acceptable level (both are successes for the reference, solver.py:407-412)."""
    status = np.asarray(status)
    return (status == OH_STATUS_CONVERGED) | (status == OH_STATUS_ACCEPTABLE)


def worse_status(a, b) -> np.ndarray:
    """Status of a problem made of independent parts (dual_arm.py: one handle per arm): the graver of the two."""
    a, b = np.asarray(a), np.asarray(b)
    return np.where(_SEVERITY[a] >= _SEVERITY[b], a, b).astype(np.int32)
OH_PROBLEM_KINEMATICS = 0
OH_PROBLEM_FIGURE_EIGHT = 1
OH_PROBLEM_POINT_MASS_MPC = 2
OH_PROBLEM_IK = 3
OH_PROBLEM_QP = 4
OH_PROBLEM_TAPE = 5
OH_PROBLEM_TORQUE_MPC = 6
OH_HESSIAN_GAUSS_NEWTON, OH_HESSIAN_EXACT, OH_HESSIAN_HYBRID = 0, 1, 2


class oh_chain(C.Structure):
    _fields_ = [
        ("ndof", C.c_int),
        ("n_chain", C.c_int),
        ("jtype", C.c_int * OH_MAX_CHAIN),
        ("qidx", C.c_int * OH_MAX_CHAIN),
        ("axcode", C.c_int * OH_MAX_CHAIN),
        ("r0ident", C.c_int * OH_MAX_CHAIN),
        ("R0", (C.c_double * 9) * OH_MAX_CHAIN),
        ("p0", (C.c_double * 3) * OH_MAX_CHAIN),
        ("axis", (C.c_double * 3) * OH_MAX_CHAIN),
        ("quat0", (C.c_double * 4) * OH_MAX_CHAIN),
        ("R_tool", C.c_double * 9),
        ("p_tool", C.c_double * 3),
        ("quat_tool", C.c_double * 4),
        ("has_lead", C.c_int),
        ("lead_axcode", C.c_int),
        ("lead_R0", C.c_double * 9),
        ("lead_p0", C.c_double * 3),
        ("lead_axis", C.c_double * 3),
    ]


OH_MAX_BODIES = 10


class oh_dynamics(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("ndof", C.c_int),
        ("R0", (C.c_double * 9) * OH_MAX_BODIES),
        ("xyz", (C.c_double * 3) * OH_MAX_BODIES),
        ("axis", (C.c_double * 3) * OH_MAX_BODIES),
        ("mass", C.c_double * OH_MAX_BODIES),
        ("com", (C.c_double * 3) * OH_MAX_BODIES),
        ("inertia", (C.c_double * 9) * OH_MAX_BODIES),
        ("vd0", C.c_double * 3),
    ]


class oh_problem_desc(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("T", C.c_int),
        ("ndof", C.c_int),
        ("dt", C.c_double),
        ("w_path", C.c_double),
        ("w_vel", C.c_double),
        ("local_path", C.POINTER(C.c_double)),
        ("lock_orientation", C.c_int),
        ("fix_dq0", C.c_int),
        ("path_in_frame", C.c_int),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("tol_feas", C.c_double),
        ("hessian", C.c_int),
        ("mu0", C.c_double),
    ]


class oh_pointmass_desc(C.Structure):
    _fields_ = [
        ("T", C.c_int),
        ("dt", C.c_double),
        ("w_acc", C.c_double),
        ("ylim", C.c_double),
        ("vlim", C.c_double),
        ("safe", C.c_double),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("track_final_only", C.c_int),
        ("w_vel", C.c_double),
        ("fix_final_velocity", C.c_int),
    ]


class oh_guards(C.Structure):
    _fields_ = [
        ("limits", C.c_int),
        ("q_lo", C.c_double * OH_MAX_CHAIN),
        ("q_up", C.c_double * OH_MAX_CHAIN),
        ("n_links", C.c_int),
        ("link_joint", C.c_int * OH_MAX_SPHERE_LINKS),
        ("link_offset", (C.c_double * 3) * OH_MAX_SPHERE_LINKS),
        ("n_obstacles", C.c_int),
        ("rho0", C.c_double),
        ("vel_limits", C.c_int),
        ("dq_lo", C.c_double * OH_MAX_CHAIN),
        ("dq_up", C.c_double * OH_MAX_CHAIN),
    ]


class oh_qp_desc(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("me", C.c_int), ("max_iter", C.c_int), ("tol", C.c_double)]


class oh_tape_desc(C.Structure):
    _fields_ = [
        ("nx", C.c_int),
        ("np", C.c_int),
        ("len", C.c_int),
        ("op", C.POINTER(C.c_int)),
        ("a", C.POINTER(C.c_int)),
        ("b", C.POINTER(C.c_int)),
        ("c", C.POINTER(C.c_double)),
        ("out_cost", C.c_int),
        ("n_ineq", C.c_int),
        ("n_eq", C.c_int),
        ("rows", C.POINTER(C.c_int)),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("tol_feas", C.c_double),
        ("rho0", C.c_double),
        ("jit", C.c_int),
        ("no_wave", C.c_int),
        ("lbfgs", C.c_int),
    ]


class oh_ik_desc(C.Structure):
    _fields_ = [
        ("ndof", C.c_int),
        ("w_nominal", C.c_double),
        ("q_lo", C.c_double * OH_MAX_CHAIN),
        ("q_up", C.c_double * OH_MAX_CHAIN),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("tol_feas", C.c_double),
        ("rho0", C.c_double),
    ]


class oh_torque_desc(C.Structure):
    _fields_ = [
        ("T", C.c_int),
        ("ndof", C.c_int),
        ("dt", C.c_double),
        ("w_path", C.c_double),
        ("w_vel", C.c_double),
        ("w_tau", C.c_double),
        ("tau_lo", C.c_double * OH_MAX_CHAIN),
        ("tau_up", C.c_double * OH_MAX_CHAIN),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("tol_compl", C.c_double),
        ("mu_barrier0", C.c_double),
        ("mu0", C.c_double),
        ("vel_limits", C.c_int),
        ("dq_lo", C.c_double * OH_MAX_CHAIN),
        ("dq_up", C.c_double * OH_MAX_CHAIN),
    ]


class OptasHipError(RuntimeError):
    pass


_LIB: Optional[C.CDLL] = None

# every symbol include/optas_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "oh_create",
    "oh_create_pointmass",
    "oh_create_ik",
    "oh_create_qp",
    "oh_qp_set_tape",
    "oh_create_tape",
    "oh_create_torque",
    "oh_tape_compile",
    "oh_tape_probe",
    "oh_tape_set_metric",
    "oh_set_constants",
    "oh_set_constants_device",
    "oh_get_constants",
    "oh_comm_unique_id",
    "oh_comm_init",
    "oh_comm_broadcast_constants",
    "oh_comm_barrier",
    "oh_comm_allreduce_max",
    "oh_comm_allreduce_sum",
    "oh_comm_allgather",
    "oh_comm_destroy",
    "oh_comm_info",
    "oh_max_batch",
    "oh_get_flag",
    "oh_set_guards",
    "oh_solve",
    "oh_solve_device",
    "oh_pm_rollout",
    "oh_tq_rollout",
    "oh_get_multipliers",
    "oh_set_dynamics",
    "oh_rnea",
    "oh_rnea_device",
    "oh_rnea_jac",
    "oh_rnea_hess",
    "oh_fk_jac",
    "oh_fk_jac_device",
    "oh_fk_jac_soa_device",
    "oh_set_profiling",
    "oh_get_timing",
    "oh_device_count",
    "oh_set_device",
    "oh_device_malloc",
    "oh_device_free",
    "oh_memcpy_h2d",
    "oh_memcpy_d2h",
    "oh_device_synchronize",
    "oh_event_timer_start",
    "oh_event_timer_stop",
    "oh_kernel_info",
    "oh_kernel_info_handle",
    "oh_specialize",
    "oh_specialize_compile",
    "oh_specialize_info",
    "oh_last_error",
    "oh_version",
    "oh_abi_version",
    "oh_set_option",
    "oh_get_option",
    "oh_destroy",
]


def library_path() -> str:
    """In-tree liboptas_hip.so; OPTAS_HIP_LIBRARY selects another build of the same sources (tuning variants: tools/gpu_tq_waves.sh)."""
    override = os.environ.get("OPTAS_HIP_LIBRARY")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboptas_hip.so")


def load() -> C.CDLL:
    """Load liboptas_hip.so (raises if it has not been built -- there is no fallback path)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise OptasHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). optas_amd has no CPU fallback."
        )
    lib = C.CDLL(path)
    # a library built from other sources (OPTAS_HIP_LIBRARY, a stale build) would misread every descriptor below -- silently
    if not hasattr(lib, "oh_abi_version"):
        raise OptasHipError(f"{path} predates oh_abi_version(): rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    lib.oh_abi_version.restype = C.c_int
    if lib.oh_abi_version() != OH_ABI_VERSION:
        raise OptasHipError(f"{path} was built for ABI version {lib.oh_abi_version()}, this binding expects {OH_ABI_VERSION}: rebuild the library")
    vp, i, dp, ip = C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.oh_create.argtypes = [C.POINTER(oh_problem_desc), C.POINTER(vp)]
    lib.oh_create_pointmass.argtypes = [C.POINTER(oh_pointmass_desc), C.POINTER(vp)]
    lib.oh_create_ik.argtypes = [C.POINTER(oh_ik_desc), C.POINTER(vp)]
    lib.oh_create_qp.argtypes = [C.POINTER(oh_qp_desc), C.POINTER(vp)]
    lib.oh_create_tape.argtypes = [C.POINTER(oh_tape_desc), C.POINTER(vp)]
    lib.oh_qp_set_tape.argtypes = [vp, C.POINTER(oh_tape_desc)]
    lib.oh_tape_compile.argtypes = [C.POINTER(oh_tape_desc), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.oh_tape_probe.argtypes = [vp, i, vp, vp, i, vp, vp, vp, vp, vp]
    lib.oh_tape_set_metric.argtypes = [vp, vp]
    lib.oh_set_constants.argtypes = [vp, C.POINTER(oh_chain)]
    lib.oh_set_constants_device.argtypes = [vp, vp, C.c_size_t]
    lib.oh_set_guards.argtypes = [vp, C.POINTER(oh_guards)]
    lib.oh_solve.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp]
    lib.oh_solve_device.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp]
    lib.oh_pm_rollout.argtypes = [vp, i, i, i, C.c_double, vp, vp, vp, vp, vp, vp]
    lib.oh_tq_rollout.argtypes = [vp, i, i, i, C.c_double, vp, vp, vp, vp, vp, vp, vp]
    lib.oh_get_multipliers.argtypes = [vp, i, vp]
    lib.oh_set_dynamics.argtypes = [vp, C.POINTER(oh_dynamics)]
    lib.oh_rnea.argtypes = [vp, i, vp, vp, vp, vp]
    lib.oh_rnea_device.argtypes = [vp, i, vp, vp, vp, vp]
    lib.oh_fk_jac.argtypes = [vp, i, vp, vp, vp]
    lib.oh_fk_jac_device.argtypes = [vp, i, vp, vp, vp]
    lib.oh_fk_jac_soa_device.argtypes = [vp, i, vp, vp, vp]
    lib.oh_set_profiling.argtypes = [vp, i]
    lib.oh_get_timing.argtypes = [vp, dp]
    lib.oh_device_count.argtypes = [ip]
    lib.oh_set_device.argtypes = [i]
    lib.oh_device_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.oh_device_free.argtypes = [vp]
    lib.oh_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    lib.oh_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    lib.oh_device_synchronize.argtypes = []
    lib.oh_specialize.argtypes = [vp]
    lib.oh_specialize_info.argtypes = [vp, dp]
    lib.oh_specialize_compile.argtypes = [C.POINTER(oh_chain), dp]
    lib.oh_kernel_info_handle.argtypes = [vp, C.c_char_p, ip]
    lib.oh_event_timer_start.argtypes = [vp]
    lib.oh_event_timer_stop.argtypes = [vp, dp]
    lib.oh_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.oh_get_option.argtypes = [vp, C.c_char_p, dp]
    lib.oh_last_error.restype = C.c_char_p
    lib.oh_version.restype = C.c_char_p
    lib.oh_destroy.argtypes = [vp]
    lib.oh_destroy.restype = None
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("oh_last_error", "oh_version", "oh_destroy"):
            fn.restype = C.c_int
    _LIB = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != OH_OK:
        msg = load().oh_last_error().decode("utf-8", "replace")
        raise OptasHipError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")


def set_option(handle, name: str, value: float) -> None:
    """oh_set_option: a scheduling / experiment knob of ONE handle (the list is in include/optas_hip.h)."""
    check(load().oh_set_option(handle, name.encode(), float(value)), f"oh_set_option({name})")


def get_option(handle, name: str) -> float:
    v = C.c_double(0.0)
    check(load().oh_get_option(handle, name.encode(), C.byref(v)), f"oh_get_option({name})")
    return v.value


def device_count() -> int:
    n = C.c_int(0)
    rc = load().oh_device_count(C.byref(n))
    return n.value if rc == OH_OK else 0


def kernel_info(name: str) -> dict:
    """Registers / scratch / LDS / resident blocks per CU of one of the library's kernels (oh_kernel_info)."""
    out = (C.c_int * 5)()
    check(load().oh_kernel_info(name.encode(), out), "oh_kernel_info")
    v, sc, lds, blk, nb = list(out)
    return {"registers_per_lane": v, "scratch_bytes_per_lane": sc, "lds_bytes_per_block": lds, "block": blk, "blocks_per_cu": nb,
            "waves_per_simd": nb * blk / 64.0 / 4.0}


def specialize_compile(chain) -> dict:
    """hiprtc compilation of the figure-eight evaluation kernels for one chain into the disk cache (oh_specialize_compile; no device needed)."""
    info = (C.c_double * 2)()
    check(load().oh_specialize_compile(C.byref(chain), info), "oh_specialize_compile")
    return {"seconds": info[0], "from_disk_cache": bool(info[1])}


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def as_f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        out = out.reshape(shape)
    return out


class DeviceBuffer:
    """A raw HBM allocation owned by Python (used by bench.py to keep inputs resident)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(load().oh_device_malloc(C.byref(p), self.nbytes), "oh_device_malloc")
        self.ptr = p

    def upload(self, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(load().oh_memcpy_h2d(self.ptr, _ptr(a), a.nbytes), "oh_memcpy_h2d")
        return self

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(load().oh_memcpy_d2h(_ptr(out), self.ptr, out.nbytes), "oh_memcpy_d2h")
        return out

    def free(self) -> None:
        if self.ptr:
            load().oh_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
