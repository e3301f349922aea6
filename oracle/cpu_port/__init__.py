"""ORACLE-SIDE CPU BASELINE (measurement infrastructure, never product): ctypes loader for oracle/cpu_port/port.hip, the
figure-eight state machine of optas_amd/csrc compiled for the host cores.  Used by bench.py's ``cpu_baseline`` leg (kind
"port", compiled code, 1..all cores) and checked against the numpy restatement in tests/.  The parity oracle is the numpy
code in oracle/*.py, not this (it shares its arithmetic with the product by construction)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "_build", "liboracle_port.so")
SRC = os.path.join(HERE, "port.hip")
_LIB = None


def build(force: bool = False) -> str:
    csrc = os.path.join(HERE, "..", "..", "optas_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in ("oh_figure8_units.h", "oh_figure8.h", "oh_device.h", "oh_types.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -mfma/-mavx2: fma() must map to the hardware instruction on the host as it does on the device (x86-64-v3: any EPYC)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Xarch_host", "-mfma", "-Xarch_host", "-mavx2", "-shared", "-fPIC",
           "-I", os.path.join(HERE, "..", "..", "include"), "-I", csrc, "-o", OUT, SRC, "-lpthread"]
    subprocess.run(cmd, check=True)
    return OUT


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(OUT):
            raise RuntimeError(f"{OUT} not built: run __graft_entry__.build()")
        _LIB = C.CDLL(OUT)
    return _LIB


def solve(chain, T, dt, local_path, x0, p, w_path=1000.0, w_vel=0.01, max_iter=300, tol=1e-6, tol_feas=1e-9, hessian=2, threads=1):
    """Same arguments and outputs as optas_amd.backend.FigureEightBackend.solve (orientation-locked family)."""
    from optas_amd import _lib  # ctypes struct layouts of include/optas_hip.h

    lib = load()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    B = x0.shape[0]
    lp = np.ascontiguousarray(local_path, dtype=np.float64).reshape(T, 3)
    desc = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_FIGURE_EIGHT, T=T, ndof=int(chain.ndof), dt=float(dt), w_path=float(w_path), w_vel=float(w_vel),
                                local_path=lp.ctypes.data_as(C.POINTER(C.c_double)), lock_orientation=1, fix_dq0=1, path_in_frame=1, max_iter=int(max_iter),
                                tol=float(tol), tol_feas=float(tol_feas), hessian=int(hessian), mu0=0.0)
    x = np.empty_like(x0)
    f = np.empty(B)
    kkt = np.empty((B, 3))
    iters = np.empty(B, dtype=np.int32)
    status = np.empty(B, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.oh_port_solve(C.byref(desc), C.byref(chain), B, vp(x0), vp(p), vp(x), vp(f), vp(kkt), vp(iters), vp(status), int(threads))
    if rc:
        raise RuntimeError("oh_port_solve: bad arguments")
    return x, f, kkt, iters, status
