"""Round-4 probe of the torque family's interior-point state machine on the GPU: second-derivative kernel vs oracle, iteration counts and optimum vs the
numpy port (oracle/torque_ipm.py) and vs the augmented-Lagrangian port (oracle/torque.py), batch timings.  python tools/gpu_tq_ipm_probe.py [B ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

SEED = 20260927 + 5


def instances(med7, B, T=30, dt=0.1, link="lbr_link_ee"):
    rng = np.random.default_rng(SEED)
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    ts = np.arange(T) * dt
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
    qc = np.atleast_2d(qn + (rng.uniform(-0.1, 0.1, (B, 7)) if B > 1 else 0.0))
    pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
    x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
    Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
    goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
    p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
    x0 = np.zeros((B, 4 * 7 * T))
    x0[:, : 7 * T] = np.tile(qc, (1, T))
    return qc, goal, x0, p


def main():
    out = {}
    med7 = RobotModel.builtin("med7")
    # 1. second derivatives
    from oracle.robot import OracleRobot
    from oracle.torque import RneaTables, TorqueProblem, rnea_ctau_hessian, solve_torque_lm
    from oracle.torque_ipm import solve_torque_ipm

    orc = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json"))
    tb = RneaTables(orc)
    rng = np.random.default_rng(1)
    q, qd, qdd, c = (rng.normal(size=(64, 7)) for _ in range(4))
    Hg = med7.rnea_hessian(q.T, qd.T, qdd.T, c.T)
    Ho = rnea_ctau_hessian(tb, q, qd, qdd, c)
    out["hessian_max_abs_diff"] = float(np.abs(Hg - Ho).max())
    out["hessian_scale"] = float(np.abs(Ho).max())
    out["hessian_asym"] = float(np.abs(Hg - np.swapaxes(Hg, 1, 2)).max())
    print(json.dumps(out), flush=True)
    # 2. small batch vs the ports
    T = 30
    Bs = [int(a) for a in sys.argv[1:]] or [64, 1024, 8192, 1]
    prob = TorqueProblem(orc, "lbr_link_ee", T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=58.0)
    for B in Bs:
        qc, goal, x0, p = instances(med7, B)
        be = TorqueBackend(med7.kinematic_chain("lbr_link_ee"), med7.dynamics_tables(), T=T, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0,
                           max_iter=600)
        r = be.solve(x0, p)
        t0 = time.perf_counter()
        r = be.solve(x0, p)
        wall = time.perf_counter() - t0
        tm = be.timing()
        it, st = np.asarray(r.iters), np.asarray(r.status)
        rec = {"B": B, "device_ms": tm["solve_ms"], "wall_ms": wall * 1e3, "launched": tm["iterations_launched"], "converged": float((st == 0).mean()),
               "it_p50": float(np.median(it)), "it_p90": float(np.percentile(it, 90)), "it_max": int(it.max()), "stat_max": float(np.asarray(r.kkt)[:, 0].max()),
               "viol_max": float(np.asarray(r.kkt)[:, 1].max()), "cmpl_max": float(np.asarray(r.kkt)[:, 2].max())}
        if B == Bs[0]:
            cmp_ = []
            for b in range(min(B, 8)):
                o = solve_torque_ipm(prob, qc[b], np.zeros(7), goal[b], max_iter=600)
                cmp_.append({"b": b, "gpu_it": int(it[b]), "port_it": o["iters"], "gpu_f": float(r.f[b]), "port_f": o["f"],
                             "dU": float(np.abs(r.x[b, 2 * 7 * T : 3 * 7 * T].reshape(T, 7) - o["U"]).max())})
            rec["vs_port"] = cmp_
            o = solve_torque_lm(prob, qc[0], np.zeros(7), goal[0], max_iter=600)
            rec["al_port_f0"] = o["f"]
        print(json.dumps(rec), flush=True)
        out[f"B{B}"] = rec
        be.close()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tq_ipm_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
