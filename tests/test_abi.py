"""The C-ABI library loads and exports every symbol include/optas_hip.h declares; without a GPU the
compute entry points fail loudly (no CPU path).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from optas_amd import _lib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "optas_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oh_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"liboptas_hip.so does not export {name}"
    assert lib.oh_version().startswith(b"optas_hip")
    assert lib.oh_abi_version() == _lib.OH_ABI_VERSION  # the binding refuses a library built for other struct layouts (_lib.load)
    assert lib.oh_set_option(None, b"tail_threshold", C.c_double(0.0)) == 1 and lib.oh_get_option(None, b"tail_threshold", None) == 1


def test_status_helpers():
    import numpy as np

    assert list(_lib.status_ok([0, 1, 2, 3, 4])) == [True, False, False, False, True]  # converged and acceptable level are successes (solver.py:407-412)
    assert list(_lib.worse_status([0, 4, 1, 3, 0], [4, 0, 3, 1, 2])) == [4, 4, 3, 3, 2]
    assert _lib.STATUS_NAMES[3] == "Infeasible_Problem_Detected" and _lib.STATUS_NAMES[4] == "Solved_To_Acceptable_Level"


def test_struct_layout_matches_header():
    # sizeof(oh_chain): 2 ints + 4*16 ints + 16*(9+3+3+4) doubles + (9+3+4) doubles
    assert C.sizeof(_lib.oh_chain) == 8 + 256 + 8 * (16 * 19 + 16) + 8 + 8 * 15 == 2952  # + has_lead, lead_axcode, lead_R0, lead_p0, lead_axis


def test_invalid_arguments_are_rejected_without_a_device_call():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.oh_create(None, C.byref(h)) == 1  # OH_ERR_INVALID
    assert b"null" in lib.oh_last_error()
    d = _lib.oh_problem_desc(kind=99)
    assert lib.oh_create(C.byref(d), C.byref(h)) == 1
    d = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_FIGURE_EIGHT, T=2, ndof=7)
    assert lib.oh_create(C.byref(d), C.byref(h)) == 1 and b"T must be" in lib.oh_last_error()
    assert lib.oh_solve(None, 1, None, None, None, None, None, None, None) == 1
    assert lib.oh_fk_jac(None, 1, None, None, None) == 1


@pytest.mark.skipif(_lib.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.oh_problem_desc(kind=_lib.OH_PROBLEM_KINEMATICS, ndof=7)
    rc = lib.oh_create(C.byref(d), C.byref(h))
    assert rc == 2 and b"no HIP device" in lib.oh_last_error()  # OH_ERR_HIP
    import optas_amd

    with pytest.raises(_lib.OptasHipError):
        optas_amd.RobotModel.builtin("kuka_lwr").get_global_link_position("end_effector_ball", [0.0] * 7)
