import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from examples.simple_joint_space_planner import setup_solver
g = np.load("/root/repo/tests/golden/planner_golden.npz")
for T in (20, 60, 120):
    robot, solver = setup_solver(T=T, solver_options={"max_iter": 2000000})
    name = robot.get_name()
    P = g["p"][:2]
    solver.reset_parameters_batch({"nominal_joint_state": P[:, :7], "current_joint_state": P[:, 7:14], "position_goal": P[:, 14:17], "orientation_goal": P[:, 17:]})
    solver.reset_initial_seed_batch({f"{name}/q/x": np.stack([np.tile(g["q0"].reshape(-1, 1), (1, T))] * len(P))})
    sols = solver.solve_batch(); sols = solver.solve_batch()
    st = solver.stats(); be, o = solver.backend, solver.opt
    x = o.decision_variables.dict2vec(sols[0])
    print("T", T, "nx", o.nx, "free", be.inner.nx, "ok", st["success"], "evals", st["iterations"].tolist(), "ms %.1f" % be.solve_ms(), "f", st["f"].round(6).tolist(),
          "rows", float(np.abs(o.a(x, P[0])).max()), float(np.abs(o.h(x, P[0])).max()), float(o.g(x, P[0]).min()), "wave", be.flag("tape_wave"), "regs_lds", be.flag("tape_regs_lds"), flush=True)
    be.close()
