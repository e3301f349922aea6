"""CPU legs of BASELINE configs 1, 3, 4, 5 ("the reference's CPU path timed beside the GPU"): the numpy ports of the device state
machines (oracle/ik_al.py, oracle/pointmass_ipm.py, oracle/guarded.py, oracle/torque_ipm.py -- the parity oracles of those families) on the
instances the GPU leg just solved, on one process and on all usable cores.  kind = "numpy port": the reference's own solver (IPOPT through
CasADi) is not installable here (import casadi fails), so this is the same algorithm as the GPU's at numpy speed, not IPOPT's speed.

Measurement infrastructure (it imports oracle/): run by bench.py's configs block (tools/bench_configs.py) as a SUBPROCESS -- no HIP runtime in
this process, so it can fork a pool freely --

    python tools/cpu_legs.py <instances.npz> <config> <cores> <seconds>

and prints one JSON object.  `seconds` bounds the work: the one-process leg solves instances until ~seconds/3 have passed (at least 2), the
all-core leg then solves as many as fit ~2/3 seconds at the measured rate (at least one per core), all drawn in order from the file.
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
R = os.path.join(ROOT, "optas_amd", "robots")
_CTX = {}


def _init(config, path):
    """Per process: the problem constants of a config (robot tables, chains), built once."""
    from oracle.robot import OracleRobot

    d = dict(np.load(path, allow_pickle=False))
    _CTX["d"] = d
    if config == "ik":
        from oracle.structured import FoldedChain

        rob = OracleRobot(os.path.join(R, "kuka_lwr.kin.json"))
        _CTX["ch"] = FoldedChain(rob, "end_effector_ball")
    elif config == "pm":
        from oracle.problems import PointMassMPCNLP

        _CTX["nlp"] = PointMassMPCNLP()
    elif config == "guarded":
        from oracle.structured import FoldedChain

        rob = OracleRobot(os.path.join(R, "kuka_lwr.kin.json"), name="kukal")
        rob.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        _CTX["rob"] = rob
        _CTX["ch"] = FoldedChain(rob, "end_effector_ball")
    elif config == "torque":
        from oracle.torque import TorqueProblem

        _CTX["prob"] = TorqueProblem(OracleRobot(os.path.join(R, "med7.kin.json")), "lbr_link_ee", T=int(d["T"]), dt=float(d["dt"]), w_path=1000.0, w_vel=0.1, w_tau=1e-4,
                                     tau_lim=float(d["lim"]))
    _CTX["config"] = config


def _solve(i):
    """One instance of the config through its numpy port; returns (status, steps, objective)."""
    c, d = _CTX["config"], _CTX["d"]
    if c == "ik":
        from oracle.ik_al import solve_ik_al

        r = solve_ik_al(_CTX["ch"], d["qn"][i], d["qn"][i], d["pg"][i], d["lo"], d["up"], tol=1e-6, tol_feas=1e-9, max_iter=300)
        return int(r["status"]), int(r["iterations"]), float(r["f"])
    if c == "pm":
        from oracle.pointmass_ipm import solve_pointmass_ipm

        nlp, p = _CTX["nlp"], d["P"][i]
        r = solve_pointmass_ipm(20, 0.05, nlp.w, 1.5, 1.0, nlp.safe_sq, p[:2], p[2:4], p[4:44].reshape(20, 2).T, p[44:84].reshape(20, 2).T, tol=1e-8)
        return int(r["status"]), int(r["iters"]), float(r["f"])
    if c == "guarded":
        from oracle.guarded import Guards, solve_free_al

        rob, p, T = _CTX["rob"], d["p"][i], int(d["T"])
        G = Guards(lo=rob.lower_actuated_joint_limits, up=rob.upper_actuated_joint_limits, links=[str(s) for s in d["links"]], link_radii=p[7:11],
                   obs_pos=p[11:].reshape(6, 4)[:, :3], obs_radii=p[11:].reshape(6, 4)[:, 3])
        r = solve_free_al(_CTX["ch"], T, float(d["dt"]), d["offsets"], p[:7], G, Q0=np.tile(p[:7], (T, 1)), rho0=10.0, exact=False, max_iter=400)
        return int(r["status"]), int(r["iters"]), float(r["f"])
    if c == "torque":
        from oracle.torque_ipm import solve_torque_ipm

        r = solve_torque_ipm(_CTX["prob"], d["qc"][i], np.zeros(7), d["goal"][i], max_iter=600)
        return int(r["status"]), int(r["iters"]), float(r["f"])
    raise ValueError(c)


def main():
    path, config, cores, seconds = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    _init(config, path)
    n_avail = int(_CTX["d"]["n"])
    # one process
    t0 = time.perf_counter()
    res1 = []
    while (len(res1) < 2 or time.perf_counter() - t0 < seconds / 3.0) and len(res1) < n_avail:
        res1.append(_solve(len(res1)))
    t1 = time.perf_counter() - t0
    rate1 = len(res1) / t1
    # all cores: a pool of forked workers over the first n_all instances
    n_all = int(min(n_avail, max(cores, rate1 * cores * seconds * 2.0 / 3.0)))
    with mp.get_context("fork").Pool(cores, initializer=_init, initargs=(config, path)) as pool:
        pool.map(_solve, range(min(cores, n_all)))  # every worker has imported its port before the clock starts
        t0 = time.perf_counter()
        resn = pool.map(_solve, range(n_all), chunksize=max(1, n_all // (4 * cores)))
        tn = time.perf_counter() - t0
    st = np.array([r[0] for r in resn])
    it = np.array([r[1] for r in resn])
    print(json.dumps({
        "value": n_all / tn, "unit": "solves/s", "cores": cores, "kind": "numpy port",
        "sample": f"the first {n_all} instances of the GPU leg's batch on {cores} processes in {tn:.2f} s ({ {'ik': 'oracle/ik_al.py:solve_ik_al', 'pm': 'oracle/pointmass_ipm.py:solve_pointmass_ipm', 'guarded': 'oracle/guarded.py:solve_free_al', 'torque': 'oracle/torque_ipm.py:solve_torque_ipm'}[config]}: "
                  f"the numpy port of the device state machine, the family's parity oracle), mean {it.mean():.1f} steps, converged {float((st == 0).mean()):.3f}",
        "value_1core": rate1, "sample_1core": f"{len(res1)} instances on 1 process in {t1:.2f} s",
        "f_first": [r[2] for r in resn[:8]],
        "reference_solver": "IPOPT unavailable: import casadi fails (never substituted; see cpu_baseline of the headline)",
    }))


if __name__ == "__main__":
    main()
