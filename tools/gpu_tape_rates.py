"""Rates of the generic tape family: config 1's IK lowered to a tape (65 536 instances, resident), and the reference's
simple_joint_space_planner.py (280 variables, limited-memory BFGS) through HIPSolver: evaluations per solve and wall time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")  # (round 4: the planner runs one wavefront per instance; batches of perturbed problems added)
sys.path.insert(0, ROOT)
from examples.example import setup_solver as ik_setup  # noqa: E402
from examples.simple_joint_space_planner import setup_solver as planner_setup  # noqa: E402
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import TapeBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402
from optas_amd.tape import compile_problem  # noqa: E402

kuka = RobotModel.builtin("kuka_lwr")
tp = compile_problem(ik_setup(build_only=True)[1])
rng = np.random.default_rng(20260927)
B = 65536
qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits,
                                                                         kuka.upper_actuated_joint_limits).T)).T
p = np.ascontiguousarray(np.concatenate([qn, pg], 1))
be = TapeBackend(tp, max_iter=2000)
bufs = [_lib.DeviceBuffer(a.nbytes) for a in (qn, p)]
bufs[0].upload(np.ascontiguousarray(qn))
bufs[1].upload(p)
d = [_lib.DeviceBuffer(qn.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B), _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)]
ms = []
for _ in range(4):
    be.solve_device(B, bufs[0], bufs[1], *d)
    ms.append(be.solve_ms())
st, it, kkt = d[4].download(np.int32, (B,)), d[3].download(np.int32, (B,)), d[2].download(np.float64, (B, 3))
out = {"tape_ik": {"config": "1 example.py IK lowered to the generic tape family (OH_PROBLEM_TAPE)", "batch": B, "tape_instructions": int(len(tp.op)),
                   "device_ms": float(np.median(ms[1:])), "solves_per_s": B / float(np.median(ms[1:])) * 1e3, "converged_frac": float((st == 0).mean()),
                   "evals_p50": float(np.median(it)), "evals_p90": float(np.percentile(it, 90)), "evals_max": int(it.max()),
                   "stationarity_max_converged": float(kkt[st == 0, 0].max()), "feasibility_max_converged": float(kkt[st == 0, 1].max())}}
g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
robot, solver = planner_setup(solver_options={"max_iter": 400000})
name = robot.get_name()
P = g["p"]
nb = len(P)
solver.reset_parameters_batch({"nominal_joint_state": P[:, :7], "current_joint_state": P[:, 7:14], "position_goal": P[:, 14:17], "orientation_goal": P[:, 17:]})
solver.reset_initial_seed_batch({f"{name}/q/x": np.stack([np.tile(g["q0"].reshape(-1, 1), (1, 20))] * nb)})
solver.solve_batch()
t0 = time.perf_counter()
solver.solve_batch()
wall = time.perf_counter() - t0
stt = solver.stats()
out["planner"] = {"config": "simple_joint_space_planner.py (280 variables, 154 + 40 rows), limited-memory BFGS", "instances": nb, "wall_ms": wall * 1e3,
                  "evals": [int(v) for v in np.atleast_1d(stt["iterations"])], "f_rel_diff_to_golden": [float(abs(a - b) / b) for a, b in zip(np.atleast_1d(stt["f"]), g["f"])],
                  "success": bool(stt["success"])}
out["planner"]["device_ms"] = float(solver.backend.solve_ms()) if hasattr(solver.backend, "solve_ms") else None
out["planner"]["path"] = {k: solver.backend.flag(k) for k in ("tape_wave", "tape_levels", "tape_passes")}
out["planner"]["us_per_evaluation_slowest_instance"] = out["planner"]["device_ms"] * 1e3 / max(out["planner"]["evals"]) if out["planner"]["device_ms"] else None
# batches of perturbed planner problems (joint states +-0.05 rad, goal +-2 cm around the golden instances), straight through the backend
be2 = solver.backend
x0 = np.zeros((nb, solver.opt.nx))
x0[:, :140] = np.tile(g["q0"].reshape(-1, 1), (1, 20)).reshape(-1)[None, :]
for Bn in (64, 256, 1024, 4096):
    idx = np.arange(Bn) % nb
    Pn = P[idx].copy()
    Pn[:, :14] += rng.uniform(-0.05, 0.05, (Bn, 14))
    Pn[:, 14:17] += rng.uniform(-0.02, 0.02, (Bn, 3))
    r = be2.solve(np.ascontiguousarray(x0[idx]), np.ascontiguousarray(Pn))
    ms_ = float(be2.solve_ms())
    it_ = np.asarray(r.iters)
    out["planner_b%d" % Bn] = {"batch": Bn, "device_ms": ms_, "solves_per_s": Bn / ms_ * 1e3, "converged_frac": float((np.asarray(r.status) == 0).mean()),
                               "evals_p50": float(np.median(it_)), "evals_max": int(it_.max()), "evaluations_per_s": float(it_.sum() / ms_ * 1e3)}
print(json.dumps(out))
