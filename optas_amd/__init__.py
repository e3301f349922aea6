"""optas_amd -- MI355X-native batched NLP solver backend behind the optas.solver.Solver interface.

Host side: plain Python + ctypes over liboptas_hip.so (hand-written HIP for gfx950).  No torch.
"""
from .spatialmath import *  # noqa: F401,F403  (the reference re-exports its spatialmath, optas/__init__.py:3)
from .models import RobotModel, TaskModel, Model, JointTypeNotSupported  # noqa: F401
from . import _lib  # noqa: F401
from .builder import OptimizationBuilder  # noqa: F401,E402  (optas/__init__.py:5)
from .solver import HIPSolver, Solver  # noqa: F401,E402  (optas/__init__.py:6 exports its solver classes)
from .expr import atan2, horzcat, path_in_frame, sumsqr, transpose, vertcat  # noqa: F401,E402  (the casadi functions the scripts use, optas/__init__.py:2)

import numpy as np


def diag(v):
    """casadi.diag of a list of numbers (``optas.diag([1e3, 1e3, 1e3])``, example/torque_control_example.py:84): the diagonal matrix."""
    return np.diag(np.asarray(v, dtype=np.float64).reshape(-1))


def deg2rad(x):
    """optas/__init__.py:10-17."""
    return (np.pi / 180.0) * np.asarray(x, dtype=np.float64)


def rad2deg(x):
    """optas/__init__.py:20-27."""
    return (180.0 / np.pi) * np.asarray(x, dtype=np.float64)


__version__ = "0.1.0"
