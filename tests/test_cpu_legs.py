"""tools/cpu_legs.py -- the CPU legs of BASELINE configs 1, 3, 4, 5 in the bench line's `configs` block (numpy ports on host cores, a subprocess without HIP) -- on small
instance files made here with the oracle alone: the tool runs, solves every instance to convergence on one process and on a pool, and reports the fields
tools/bench_configs.py reads."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import KUKA_KIN, MED7_KIN
from oracle.robot import OracleRobot
from oracle.structured import FoldedChain
from oracle.torque import TorqueProblem

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def _files(tmp_path):
    rng = np.random.default_rng(1)
    n = 6
    rob = OracleRobot(KUKA_KIN)
    ch = FoldedChain(rob, "end_effector_ball")
    lo, up = rob.lower_actuated_joint_limits, rob.upper_actuated_joint_limits
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (n, 7))
    pg = ch.fk(np.clip(qn + rng.uniform(-0.5, 0.5, (n, 7)), lo, up))[0]
    np.savez(tmp_path / "ik.npz", n=n, qn=qn, pg=pg, lo=lo, up=up)
    from examples.point_mass_mpc import obstacle_and_goal

    obs, _ = obstacle_and_goal(2.0, np.zeros(2))
    P = []
    while len(P) < n:
        c = rng.uniform(-1.2, 1.2, 2)
        if np.linalg.norm(c - obs[:, 0]) <= 0.35:
            continue
        goal = np.stack([np.clip(c[j] + (1 - c[j]) * np.arange(20) / 19.0, -1.5, 1.5) for j in range(2)])
        P.append(np.concatenate([c, np.zeros(2), goal.T.reshape(-1), obs.T.reshape(-1)]))
    np.savez(tmp_path / "pm.npz", n=n, P=np.array(P))
    prob = TorqueProblem(OracleRobot(MED7_KIN), "lbr_link_ee", T=12, dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=58.0)
    qc = np.deg2rad([0, 30, 0, -90, 0, -30, 0]) + rng.uniform(-0.05, 0.05, (3, 7))
    np.savez(tmp_path / "torque.npz", n=3, qc=qc, goal=np.stack([prob.goal_figure_eight(q) for q in qc]), T=12, dt=0.1, lim=58.0)
    from examples.dual_arm import SPHERE_LINKS, path_offsets

    T = 30
    offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    qa = np.deg2rad([0, -30, 0, 90, 0, 30, 0]) + rng.uniform(-0.02, 0.02, (3, 7))
    obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
    p = np.concatenate([qa, np.full((3, 4), 0.1), np.tile(obs_row, (3, 1))], 1)
    np.savez(tmp_path / "guarded.npz", n=3, p=p, T=T, dt=10.0 / (T - 1), offsets=offs.T, links=np.array(SPHERE_LINKS))


@pytest.mark.parametrize("config", ["ik", "pm", "guarded", "torque"])
def test_cpu_leg_runs_and_reports(tmp_path, config):
    _files(tmp_path)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_legs.py"), str(tmp_path / f"{config}.npz"), config, "2", "1.5"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    leg = json.loads(cp.stdout.strip().splitlines()[-1])
    assert leg["kind"] == "numpy port" and leg["unit"] == "solves/s" and leg["cores"] == 2 and leg["value"] > 0 and leg["value_1core"] > 0
    assert "converged 1.000" in leg["sample"] and len(leg["f_first"]) >= 2 and all(np.isfinite(leg["f_first"]))
    assert "IPOPT unavailable" in leg["reference_solver"]
