"""The planner at longer horizons on the wavefront-per-instance tape path: T = 20 (280 variables; registers fit the LDS), T = 60 (840 variables),
T = 120 (1680 variables; the register file lives in global memory).  Prints (JSON; profiles/r04_tape_horizons.json) set-up time, registers, placement, evaluations, device time, rows of
the literal problem at the answer.  python tools/gpu_tape_big.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from examples.simple_joint_space_planner import setup_solver
g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
out = []
for T in (20, 60, 120):
    t0 = time.time()
    robot, solver = setup_solver(T=T, solver_options={"max_iter": 2000000})
    name = robot.get_name()
    set_up = time.time() - t0
    P = g["p"][:2]
    solver.reset_parameters_batch({"nominal_joint_state": P[:, :7], "current_joint_state": P[:, 7:14], "position_goal": P[:, 14:17], "orientation_goal": P[:, 17:]})
    solver.reset_initial_seed_batch({f"{name}/q/x": np.stack([np.tile(g["q0"].reshape(-1, 1), (1, T))] * len(P))})
    t0 = time.time()
    sols = solver.solve_batch()
    wall = time.time() - t0
    st = solver.stats()
    be, o = solver.backend, solver.opt
    rows = []
    for b in range(len(P)):
        x = o.decision_variables.dict2vec(sols[b])
        rows.append((float(np.abs(o.a(x, P[b])).max()), float(np.abs(o.h(x, P[b])).max()), float(o.g(x, P[b]).min())))
    out.append({"T": T, "nx": int(o.nx), "tape_len": len(be.tape.op), "set_up_s": round(set_up, 2), "wall_s": round(wall, 3), "status": np.asarray(st["status"]).tolist(),
           "f": [round(float(v), 6) for v in st["f"]], "evaluations": np.asarray(st["iterations"]).tolist() if "iterations" in st else None,
           "flags": {k: be.flag(k) for k in ("tape_wave", "tape_regs_lds", "tape_levels", "tape_passes")}, "rows_max_abs_a__max_abs_h__min_g": rows})
    print(out[-1], file=sys.stderr, flush=True)
print(json.dumps({"what": "simple_joint_space_planner.py at T = 20 / 60 / 120 on the wavefront-per-instance tape path, 2 golden parameter sets each", "runs": out}))
