#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: MPC solves/sec, KUKA LWR 7-DoF figure-eight, T=50.

  python bench.py --gpus 1 --steps 5 --warmup 1            (driver: N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            (driver: N>1, one rank per GPU)

A "step" is one pass of the hot path over one batch: one batched NLP solve (oh_solve_device) of B
independent instances per GPU, inputs (seeds x0, parameters qc) already resident in HBM.  Instances are
SURVEY 8(d)'s synthetic set: qc = deg2rad[0,30,0,-90,0,-30,0] + U(-0.1,0.1)^7, seed = qc repeated,
numpy default_rng(20260927 + rank).  Multi-GPU: instances shard across ranks with no data-path
collective; rank 0 alone sets the kinematic constants (oh_chain, 2952 B) and the library broadcasts them
once over RCCL/xGMI (oh_comm_broadcast_constants).  No torch anywhere: the communicator lives inside
liboptas_hip, Python only carries the 128-byte RCCL unique id from rank 0 to the others through a file
(optas_amd/distributed.py); barriers and the MAX-reduce of the elapsed time go through oh_comm_* too.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# code objects of the run-time specialised kernels (oh_specialize): kept inside the tree, so a cache filled by __graft_entry__.build() travels
os.environ.setdefault("OPTAS_HIP_CACHE", os.path.join(ROOT, ".optas_hip_cache"))

import optas_amd  # noqa: E402
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import FigureEightBackend  # noqa: E402

T = 50
TMAX = 10.0
LINK = "end_effector_ball"
QC0_DEG = [0, 30, 0, -90, 0, -30, 0]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Algorithmic bytes per unit = (instance, free knot) per launch, f64:
#   k_eval   : (two launches: k_retract leaves the retracted trial knot in HBM, k_evalb evaluates it; timed as one region)
#              read q 7, V 18, z 4, model (e, Jp Z) 15 ; write q 7, V 18, Dr 10, g 7, phi 1, cv 1, model 15 = 103 doubles
#              (the hand-over of the retracted knot between the two launches, 7 doubles written and read again, is traffic, not
#               algorithmic bytes)
#              (V: the three Householder vectors of the null-space basis, 3N - 3 doubles; Z itself is rebuilt in registers.
#               model: end-effector position and Jp Z of the knot, what the next retraction's position target is predicted from)
#   k_couple : read V_t 18 (V_{t+1} is an L2 hit), q 3x7, g 7, phi 1 ; write E 16, gt 4, merit 1 = 68 doubles
#   k_step   : read merit 1, cv 1, E 16, Dr 10, gt 4 ; write+read gains 20+20 ; write z 4  = 76 doubles (80 until the end of round 5: the forward pass read gt again)
BYTES = {"k_eval": 103 * 8, "k_couple": 68 * 8, "k_step": 76 * 8}
# Round 3, coupling folded into evaluation and sweep (oh_get_flag "fuse_couple"; no k_couple launch):
#   k_eval   : + the two neighbours' retracted knots (2 x 7 read); G instead of g and the merit share instead of phi out (same count) = 117 doubles
#   k_step   : read merit 1, cv 1, V 18, G 7, Dr 10 ; write+read gains 20+20 ; write z 4 = 81 doubles
#              (89 until the end of round 5: the reduced gradients gt went out to the forward pass and came back, 4 + 4, for the directional derivative g.z --
#               which is -sum_t |L_t^-1 r_t|^2, a by-product of the backward substitutions)
BYTES_ZC = {"k_eval": 117 * 8, "k_couple": 0, "k_step": 81 * 8}
BYTES_FKJAC = 448  # q 56 B in, pose 56 B + J 336 B out


def make_inputs(B: int, rank: int):
    rng = np.random.default_rng(20260927 + rank)
    qc = np.deg2rad(QC0_DEG)[None, :] + rng.uniform(-0.1, 0.1, (B, 7))
    x0 = np.concatenate([np.repeat(qc, T, axis=0).reshape(B, 7 * T), np.zeros((B, 7 * (T - 1)))], axis=1)
    return x0, qc


def local_path():
    t = np.linspace(0.0, TMAX, T)
    lp = np.zeros((T, 3))
    lp[:, 0] = 0.2 * np.sin(t * np.pi * 0.5)
    lp[:, 1] = 0.1 * np.sin(t * np.pi)
    return float(t[1] - t[0]), lp


def usable_cores() -> int:
    """Host threads worth starting: the affinity mask, capped by the cgroup CPU quota (a container can see 256 cores and own 16)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(round(q / per))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(sample: int, hessian: str = "hybrid", tol: float = 1e-8):
    """The same structured SQP on this box's host cores: (i) the compiled port (oracle/cpu_port: the state machine of the HIP
    kernels built for x86, std::thread over instances) on one core and on all cores, (ii) the numpy restatement
    (oracle/structured.py, the parity oracle) on a few instances.  IPOPT, the reference's own solver, is probed and reported."""
    from oracle import cpu_port
    from oracle.robot import OracleRobot
    from oracle.structured import StructuredFigureEight, solve_structured_lm

    os.environ.setdefault("OMP_NUM_THREADS", "1")
    try:
        import casadi  # noqa: F401

        ipopt = "casadi importable (not timed: the reference's graph builder is not part of this repo)"
    except Exception as e:  # the expected case
        ipopt = f"IPOPT unavailable: import casadi fails ({type(e).__name__})"
    ncores = usable_cores()
    hmode = {"gauss_newton": 0, "exact": 1, "hybrid": 2}[hessian]
    dt, lp = local_path()
    chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(LINK)
    # one core: `sample` instances; all usable cores: as many instances as ~10 s of work at the one-core rate
    x0, qc = make_inputs(sample, 0)
    cpu_port.solve(chain, T, dt, lp, x0[:8], qc[:8], hessian=hmode, tol=tol, threads=1)  # warm-up
    t0 = time.perf_counter()
    _, _, _, it1, st1 = cpu_port.solve(chain, T, dt, lp, x0, qc, hessian=hmode, tol=tol, threads=1)
    t_1 = time.perf_counter() - t0
    nall = int(min(131072, max(4096, 10.0 * (sample / t_1) * ncores)))
    x0, qc = make_inputs(nall, 0)
    t0 = time.perf_counter()
    _, f_port_all, _, itn, stn = cpu_port.solve(chain, T, dt, lp, x0, qc, hessian=hmode, tol=tol, threads=ncores)
    t_n = time.perf_counter() - t0
    # numpy restatement, a handful of instances (it is ~50x slower than compiled code)
    robot = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    prob = StructuredFigureEight(robot, LINK, T=T, Tmax=TMAX)
    n_np = min(sample, 32)
    t0 = time.perf_counter()
    its = [solve_structured_lm(prob, qc[i], max_iter=300, tol=tol, hessian=hessian)["iters"] for i in range(n_np)]
    t_np = time.perf_counter() - t0
    # the reference's own alternative back-end: scipy SLSQP wired like ScipyMinimizeSolver (solver.py:652-679: v(x) >= 0 as one "ineq" block with
    # Jacobian dv) on the literal 693-variable / 1114-row problem.  It makes no progress on this problem (300 iterations, 7 minutes,
    # f 1225 -> 1224.65), so the leg is bounded by iterations and reports what it reached.
    from oracle.problems import FigureEightNLP
    from oracle.solvers import scipy_minimize

    nlp = FigureEightNLP(robot, LINK, T=T, Tmax=TMAX)
    n_sl = 4
    t0 = time.perf_counter()
    rs = scipy_minimize(nlp, nlp.seed(qc[0]), qc[0], method="SLSQP", tol=1e-6, options={"maxiter": n_sl})
    t_sl = time.perf_counter() - t0
    f_star = solve_structured_lm(prob, qc[0], max_iter=300, tol=tol, hessian=hessian)["f"]
    # the reference's ALGORITHM CLASS on the reference's FORM (IPOPT is what CasADiSolver.setup("ipopt") runs, solver.py:355-398): oracle/ipm_reference_form.py,
    # a primal-dual interior point with filter line search after Waechter & Biegler 2006 with IPOPT's default parameters, on the literal 693-variable /
    # 1114-row problem, from the reference's seed, one instance (dense numpy linear algebra where IPOPT has MUMPS: an algorithm baseline, not IPOPT's speed)
    from oracle.ipm_reference_form import solve_ipm
    from oracle.problems import FastFigureEightNLP

    nlpf = FastFigureEightNLP(robot, LINK, T=T, Tmax=TMAX)
    t0 = time.perf_counter()
    ri = solve_ipm(nlpf, nlpf.seed(qc[0]), qc[0], max_iter=400)
    t_ipm = time.perf_counter() - t0
    return {
        "reference_algorithm": {
            "what": "interior-point filter line search (oracle/ipm_reference_form.py: Waechter-Biegler 2006, IPOPT defaults incl. bound_relax_factor 1e-8) on the literal "
            "min f s.t. 0 <= v <= 1e10, equalities as (e, -e) pairs, exact Lagrangian Hessian, from the reference seed; first instance of the batch, numpy on the host",
            "status": ri["status"],
            "iterations": int(ri["iters"]),
            "seconds": t_ipm,
            "solves_per_s": (1.0 / t_ipm) if ri["status"] in ("optimal", "acceptable") else 0.0,
            "f_reached": float(ri["f"]),
            "f_structured_optimum": float(f_star),
            "note": "f_reached sits sum|lam| 1e-8 ~ 7e-6 below the exactly feasible optimum: IPOPT's bound relaxation lets every row of v end 1e-8 below zero",
        },
        "value": nall / t_n,
        "unit": "solves/s",
        "cores": ncores,
        "kind": "port",
        "reference_wired_scipy": {
            "what": "scipy SLSQP wired like the reference's ScipyMinimizeSolver (solver.py:652-679) on the literal layout, first instance, 1 thread",
            "iterations": int(rs.nit),
            "seconds": t_sl,
            "converged": bool(rs.success),
            "f_reached": float(rs.fun),
            "f_optimum": float(f_star),
            "solves_per_s": (1.0 / t_sl) if rs.success else 0.0,
            "note": f"stopped after {n_sl} iterations ({t_sl / max(1, rs.nit):.2f} s each): SLSQP does not converge on this problem (rank-deficient quaternion rows), see SURVEY App. D",
        },
        "sample": f"{nall} instances of the same workload (first of rank 0's batch) on {ncores} threads in {t_n:.2f} s, compiled port of the HIP state machine "
        f"(oracle/cpu_port), tol {tol:g}, mean {float(np.mean(itn)):.0f} iterations, converged {float(np.mean(stn == 0)):.4f}",
        "_f_port": f_port_all,  # (popped by main: the objectives the host port reached on the first `nall` instances of rank 0's batch, compared with the GPU's)
        "value_1core": sample / t_1,
        "sample_1core": f"{sample} instances on 1 thread in {t_1:.2f} s, mean {float(np.mean(it1)):.0f} iterations",
        "numpy_port_value": n_np / t_np,
        "numpy_port_sample": f"{n_np} instances, oracle/structured.py:solve_structured_lm (the parity oracle), 1 thread, {t_np:.1f} s, mean {float(np.mean(its)):.0f} iterations",
        "reference_solver": ipopt,
    }


def oracle_sample(x0, qc, x, f, status, n: int, tol: float = 1e-8):
    """`n` instances of the batch the timed steps solved, graded by the oracle (never by the library): reference-form KKT residuals of
    min f s.t. 0 <= v <= 1e10 on the literal 1114-row v (oracle/solvers.py:kkt_reference_form, what "KKT residual vs IPOPT" is reported on), the
    reference objective recomputed from x, and the optimum the compiled host port of the state machine reaches from the same seed."""
    from oracle import cpu_port
    from oracle.problems import FigureEightNLP
    from oracle.robot import OracleRobot
    from oracle.solvers import kkt_reference_form

    robot = OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "kuka_lwr.kin.json"))
    nlp = FigureEightNLP(robot, LINK, T=T, Tmax=TMAX)
    B = len(qc)
    idx = np.sort(np.random.default_rng(B).choice(B, min(n, B), replace=False))
    dt, lp = local_path()
    chain = optas_amd.RobotModel.builtin("kuka_lwr").kinematic_chain(LINK)
    t0 = time.perf_counter()
    ks = [kkt_reference_form(nlp, x[i], qc[i]) for i in idx]
    f_ref = np.array([nlp.f(x[i], qc[i]) for i in idx])
    _, f_port, _, _, st_port = cpu_port.solve(chain, T, dt, lp, x0[idx], qc[idx], tol=tol, threads=usable_cores())
    same = np.abs(f[idx] - f_port) <= 1e-9 * np.abs(f_port)
    return {
        "instances": [int(i) for i in idx],
        "what": "reference-form KKT (min f s.t. 0 <= v <= 1e10, literal v = [a; -a; h; -h], multipliers by bounded least squares) evaluated by oracle/ on x "
        "downloaded after the timed steps; objective recomputed by oracle/problems.py:FigureEightNLP.f; optimum of oracle/cpu_port from the same seed",
        "stationarity_max": float(max(k["stationarity"] for k in ks)),
        "feasibility_max": float(max(k["feasibility"] for k in ks)),
        "complementarity_max": float(max(k["complementarity"] for k in ks)),
        "objective_recomputed_max_abs_diff": float(np.abs(f_ref - f[idx]).max()),
        "objective_equals_host_port_1e-9": int(same.sum()),
        "objective_max_rel_diff_vs_host_port": float((np.abs(f[idx] - f_port) / np.abs(f_port)).max()),
        "all_converged": bool((status[idx] == 0).all() and (st_port == 0).all()),
        "tolerances": {"stationarity": 1e-5, "feasibility": 1e-9, "complementarity": 1e-8, "objective": 1e-9},
        "seconds": time.perf_counter() - t0,
    }


def per_rank_block(elapsed_s, device_ms_per_step, rate, reduce_max, reduce_sum, rccl_world, note):
    """What a multi-rank run reports about the spread over its ranks (the same code in the real run, over RCCL, and in --dry-run, over the rendezvous carrier)."""
    return {"elapsed_s_max": reduce_max(elapsed_s), "elapsed_s_min": -reduce_max(-elapsed_s),
            "device_ms_per_step_max": reduce_max(device_ms_per_step), "device_ms_per_step_min": -reduce_max(-device_ms_per_step),
            "sum_of_rank_rates_solves_per_s": reduce_sum(rate), "rccl_world": rccl_world, "note": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=262144, help="instances per GPU per step (SURVEY 8(d): B in {256, 4096, 65536})")
    ap.add_argument("--max-iter", type=int, default=300)
    ap.add_argument("--tol", type=float, default=1e-8, help="stopping tolerance on the reduced gradient, unscaled.  Default = IPOPT's `tol` default, what every reference "
                    "config runs with (setup('ipopt') without options: figure_eight_plan.py:111); rounds 1-5 quoted 1e-6, which the `tol_1e-6` block still reports")
    ap.add_argument("--hessian", choices=["gauss_newton", "exact", "hybrid"], default="hybrid")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="instances timed on one host core")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fk-units", type=int, default=1 << 22)
    ap.add_argument("--oracle-sample", type=int, default=16, help="instances of the timed batch graded by the oracle afterwards (rank 0, N=1)")
    ap.add_argument("--timed-only", action="store_true", help="only the timed K steps and their per-kernel pass: no small-batch latency, no PCIe-inclusive solve, no "
                    "configs block (tools/profile.sh: the rocprofv3 averages then cover exactly the launches the roofline object is computed from)")
    ap.add_argument("--no-configs", action="store_true", help="skip the block that measures BASELINE configs 1, 3, 4, 5 after the timed region (rank 0, N=1)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: run the multi-process plumbing only (rendezvous of the RCCL id with a stand-in id, "
                    "per-rank inputs) and print one JSON line per rank; everything of a --gpus N run except oh_comm_init and the solves")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        import hashlib

        from optas_amd import distributed as oad

        uid = oad.exchange(rank, world, lambda: os.urandom(_lib.OH_COMM_ID_BYTES), timeout=60.0) if "RANK" in os.environ else b""
        if "RANK" in os.environ and os.environ.get("OPTAS_RDZV", "file") == "file":
            # a record counts only while its publisher lives; in a real run rank 0 is then inside ncclCommInitRank until everyone has joined --
            # here it waits for the others' acknowledgements instead
            ack = oad.rendezvous_path() + ".ack"
            if rank == 0:
                t_ack = time.monotonic()
                while not all(os.path.exists(f"{ack}{r}") for r in range(1, world)) and time.monotonic() - t_ack < 60.0:
                    time.sleep(0.01)
                for r in range(1, world):
                    try:
                        os.remove(f"{ack}{r}")
                    except OSError:
                        pass
            else:
                open(f"{ack}{rank}", "w").close()
        x0, qc = make_inputs(min(args.batch, 64), rank)
        per_rank = None
        if "RANK" in os.environ:
            # the per-rank block of a real run with stand-in measurements (rank r: 1 + r / 100 s, 80 + r ms), reduced over files in the private rendezvous
            # directory instead of RCCL: the harness code that builds the block is the real one (per_rank_block)
            vdir = oad.rendezvous_dir()
            os.makedirs(vdir, mode=0o700, exist_ok=True)
            tag = hashlib.sha256(uid).hexdigest()[:16]
            calls = [0]

            def gather(v):
                calls[0] += 1
                mine_path = os.path.join(vdir, f"dry_{tag}_{calls[0]}_{rank}.val")
                with open(mine_path + ".tmp", "w") as fh:
                    fh.write(repr(float(v)))
                os.replace(mine_path + ".tmp", mine_path)
                vals, t_g = [], time.monotonic()
                for r in range(world):
                    pth = os.path.join(vdir, f"dry_{tag}_{calls[0]}_{r}.val")
                    while not os.path.exists(pth):
                        if time.monotonic() - t_g > 60.0:
                            raise SystemExit(f"dry run: rank {r} never published value {calls[0]}")
                        time.sleep(0.005)
                    vals.append(float(open(pth).read()))
                return vals

            per_rank = per_rank_block(1.0 + rank / 100.0, 80.0 + rank, 1000.0 * (rank + 1), lambda v: max(gather(v)), lambda v: sum(gather(v)), None,
                                      "dry run: stand-in measurements, reduced over the rendezvous directory (a real run reduces over RCCL and reports rccl_world = WORLD_SIZE)")
        print(json.dumps({"dry_run": True, "rank": rank, "world": world, "local_rank": local_rank, "rdzv": os.environ.get("OPTAS_RDZV", "file"),
                          "id_sha256": hashlib.sha256(uid).hexdigest(), "id_bytes": len(uid), "qc_first": qc[0].tolist(), "nx": int(x0.shape[1]),
                          "device_index": local_rank, "per_rank": per_rank}), flush=True)
        return
    comm = None
    rccl_error = None
    # stdout carries ONE line, the JSON: RCCL prints its version banner to stdout when a communicator is created (and native code may print later),
    # so from here on file descriptor 1 points at stderr and the result line goes to the saved descriptor at the very end
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world > 1 or "RANK" in os.environ:  # launcher-started: one rank per GPU
        from optas_amd import distributed as oad

        comm = oad.init_from_env()  # oh_set_device(local_rank) + RCCL communicator inside liboptas_hip
    elif os.environ.get("OH_BENCH_DIST_AT_1", "1") == "1":
        # plain `python bench.py`: the same path at the size there is -- a communicator of one rank, the broadcast of the constants through it
        # (round-4 verdict, Next 8).  A box without a usable librccl still measures the solves: the reason is reported, the line says rccl_world null.
        from optas_amd import distributed as oad

        try:
            comm = oad.Communicator(0, 1, local_rank)
        except Exception as e:  # noqa: BLE001
            comm, rccl_error = None, f"{type(e).__name__}: {e}"
    lib = _lib.load()
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: liboptas_hip has no CPU path")
    _lib.check(lib.oh_set_device(local_rank), "oh_set_device")

    dt, lp = local_path()
    hess = {"gauss_newton": 0, "exact": 1, "hybrid": 2}[args.hessian]
    if comm is None or rank == 0:
        robot = optas_amd.RobotModel.builtin("kuka_lwr")
        be = FigureEightBackend(robot.kinematic_chain(LINK), T, dt, lp, max_iter=args.max_iter, tol=args.tol, hessian=hess)
    else:  # the other ranks never read the URDF: they get the folded constants from rank 0
        be = FigureEightBackend(None, T, dt, lp, max_iter=args.max_iter, tol=args.tol, hessian=hess, ndof=7)
    rccl_world = None
    if comm is not None:
        comm.broadcast_constants(be.handle, root=0)  # the one collective of the whole job
        rccl_world = comm.info()[1]  # ncclCommCount: the communicator the constants travelled over spans this many ranks
        if rccl_world != world:  # a job whose collective did not span every rank would report N GPUs and measure fewer
            raise SystemExit(f"bench.py: RCCL communicator spans {rccl_world} ranks, WORLD_SIZE is {world}")

    B = args.batch
    x0, qc = make_inputs(B, rank)
    nx = x0.shape[1]
    d_x0 = _lib.DeviceBuffer(x0.nbytes).upload(x0)
    d_p = _lib.DeviceBuffer(qc.nbytes).upload(qc)
    d_x = _lib.DeviceBuffer(x0.nbytes)
    d_f = _lib.DeviceBuffer(B * 8)
    d_k = _lib.DeviceBuffer(B * 24)
    d_it = _lib.DeviceBuffer(B * 4)
    d_st = _lib.DeviceBuffer(B * 4)

    def sync_all():
        _lib.check(lib.oh_device_synchronize(), "sync")
        if comm is not None:
            comm.barrier()

    cpu = None
    if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only; before the GPU phase so that the device work sits at the end of the run
        cpu = cpu_baseline(args.cpu_sample, args.hessian, args.tol)

    streams_used = int(be.get_option("streams")) if B >= int(be.get_option("split_min")) else 1
    be.set_profiling(False)
    for _ in range(args.warmup):
        be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
    sync_all()
    t0 = time.perf_counter()
    tm = {"eval_ms": 0.0, "step_ms": 0.0, "couple_ms": 0.0, "eval_launches": 0, "step_launches": 0, "instance_launches": 0, "solve_ms": 0.0, "rejected_steps": 0, "compactions": 0, "tail_iterations": 0}
    solve_ms_plain = 0.0
    for _ in range(args.steps):
        be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
        solve_ms_plain += be.timing()["solve_ms"]
    sync_all()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if comm is not None:
        mine = elapsed
        dev = solve_ms_plain / args.steps  # this rank's own HIP-event time of one step: rank skew shows as max - min
        per_rank = per_rank_block(mine, dev, B * args.steps / mine, comm.max_over_ranks, comm.sum_over_ranks, rccl_world,
                                  "each rank's own wall time of the K steps between the two barriers, reduced through oh_comm_allreduce_{max,sum}; value uses the max")
        elapsed = per_rank["elapsed_s_max"]
    # second pass of the same K steps with one hipEventRecord after every kernel on the handle's stream: the per-kernel times behind the
    # roofline object (the timed pass above runs without them)
    be.set_profiling(True)
    t0p = time.perf_counter()
    for _ in range(args.steps):
        be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
        t = be.timing()
        for k in tm:
            tm[k] += t[k]
    _lib.check(lib.oh_device_synchronize(), "sync")
    elapsed_profiled = time.perf_counter() - t0p
    be.set_profiling(False)

    status = d_st.download(np.int32, (B,))
    iters = d_it.download(np.int32, (B,))
    kkt = d_k.download(np.float64, (B, 3))
    fvals = d_f.download(np.float64, (B,))
    conv = status == 0
    osample = None
    if world == 1 and args.oracle_sample > 0 and not args.no_cpu_baseline:
        osample = oracle_sample(x0, qc, d_x.download(np.float64, (B, nx)), fvals, status, args.oracle_sample, args.tol)

    # north-star kernel K1 (FK + geometric Jacobian), SoA, measured with HIP events on the handle's stream
    nfk = args.fk_units
    rngq = np.random.default_rng(1)
    qsoa = rngq.uniform(-2.9, 2.9, (7, nfk))
    d_q = _lib.DeviceBuffer(qsoa.nbytes).upload(qsoa)
    d_pose = _lib.DeviceBuffer(nfk * 7 * 8)
    d_J = _lib.DeviceBuffer(nfk * 42 * 8)
    be.fk_jac_soa_device(nfk, d_q, d_pose, d_J)
    fk_ms = []
    for _ in range(5):
        be.event_timer_start()
        be.fk_jac_soa_device(nfk, d_q, d_pose, d_J)
        fk_ms.append(be.event_timer_stop())
    fk_ms = float(np.mean(fk_ms))
    # the same kernel in the layout of the ABI's reference entry points (q[n][ndof], pose[n][7], J[n][6][ndof]: what
    # RobotModel.get_global_link_*_function(link, n=N) hands out), staged through LDS
    d_q.upload(np.ascontiguousarray(qsoa.T))
    be.fk_jac_device(nfk, d_q, d_pose, d_J)
    fk_ref_ms = []
    for _ in range(5):
        be.event_timer_start()
        be.fk_jac_device(nfk, d_q, d_pose, d_J)
        fk_ref_ms.append(be.event_timer_stop())
    fk_ref_ms = float(np.mean(fk_ref_ms))
    for b in (d_q, d_pose, d_J):
        b.free()

    # small-batch latency (BASELINE configs[1] literally is batch = 1): whole solve on the device, inputs resident, median of 7
    lat = {1: None, min(1024, B): None}
    for nb in (() if args.timed_only else (1, min(1024, B))):
        ms = []
        for _ in range(8):
            be.solve_device(nb, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
            ms.append(be.timing()["solve_ms"])
        lat[nb] = float(np.median(ms[1:]))
        its_nb = d_it.download(np.int32, (B,))[:nb]
        lat[f"iters_{nb}"] = float(its_nb.mean())
    # PCIe-inclusive rate: the same problem through oh_solve from pageable host buffers (what a ctypes host that keeps nothing resident pays)
    pcie = None
    if world == 1 and not args.timed_only:
        # (outputs allocated once and touched: a fresh np.empty of a gigabyte is page faults, not PCIe)
        hx, hf, hk = np.zeros((B, nx)), np.zeros(B), np.zeros((B, 3))
        hi, hs = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)

        def host_solve(nb):
            t0h = time.perf_counter()
            _lib.check(lib.oh_solve(be.handle, nb, _lib._ptr(x0), _lib._ptr(qc), _lib._ptr(hx), _lib._ptr(hf), _lib._ptr(hk), _lib._ptr(hi), _lib._ptr(hs)), "oh_solve")
            return time.perf_counter() - t0h

        pcie = {}
        for nb in sorted({min(B, 65536), B}):
            host_solve(nb)
            t_h = min(host_solve(nb) for _ in range(2))
            pcie[f"batch_{nb}"] = {"batch": nb, "wall_ms": 1e3 * t_h, "solves_per_s": nb / t_h, "converged_frac": float((hs[:nb] == 0).mean()),
                                   "frac_of_resident_rate": (nb / t_h) / (B * args.steps / elapsed) if nb == B else None}
        pcie["solves_per_s"] = pcie[f"batch_{B}"]["solves_per_s"]
        pcie["what"] = ("oh_solve with pageable numpy buffers: x0 and p up, x, f, kkt, iters, status down (2 x 5.5 KB per instance over PCIe), one call, best of two; since round 6 a batch of "
                        ">= 2 x pipe_chunk (32 768) instances goes in chunks on two lanes (handle + peer, a stream and a host thread each: csrc/oh_api.hip:solve_pipelined), one lane's "
                        "transfers under the other lane's kernels; `solves_per_s` is the rate at the bench batch")
    occupancy = {k: be.kernel_info(k) for k in (("k_retract", "k_evalb_zc", "k_step_zc", "k_tail", "k_fk_jac") if be.flag("fuse_couple") else
                                                ("k_retract", "k_evalb", "k_couple", "k_step", "k_tail", "k_fk_jac"))}
    spec_info = be.specialize_info()

    if rank != 0:
        if comm is not None:
            comm.barrier()
            comm.destroy()
        return

    # (instance, knot) units the batched kernels actually processed (k_step counts running instances per launch;
    # the persistent tail kernel adds one per iteration per instance, but its time is not in the kernel brackets)
    units = tm["instance_launches"] * (T - 2)
    zc = bool(be.flag("fuse_couple"))
    bytes_k = BYTES_ZC if zc else BYTES
    kms = {"k_eval": tm["eval_ms"], "k_couple": 0.0 if zc else tm["couple_ms"], "k_step": tm["step_ms"] + (tm["couple_ms"] if zc else 0.0)}
    dom = max(kms, key=kms.get)
    launches = max(1, tm["step_launches"])
    per_kernel = {
        k: {"total_ms": v, "avg_launch_ms": v / launches, "bytes_per_unit": bytes_k[k], "achieved_GBps": units * bytes_k[k] / (v * 1e-3) / 1e9 if v > 0 else 0.0}
        for k, v in kms.items()
    }
    achieved = per_kernel[dom]["achieved_GBps"]
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure comes from the committed
    # rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by tools/summarize_profile.py together with the units per launch it was
    # measured at) and is rescaled to this run's units per launch; traffic_source says where it came from.
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            ent = tj.get(dom, {})
            meta = tj.get("_meta", {})
            upl = ent.get("units_per_launch") or meta.get("units_per_launch")
            if ent.get("bytes_per_launch") and upl:
                traffic = ent["bytes_per_launch"] / upl * (units / launches)
                traffic_source = {"file": "profiles/pmc_traffic.json", "profile": meta.get("tag"), "commit": meta.get("commit"), "units_per_launch_profiled": upl,
                                  "bytes_per_unit_profiled": ent["bytes_per_launch"] / upl, "rescaled_to_units_per_launch": units / launches}
        except Exception:
            traffic, traffic_source = None, None
    roofline = {
        "kernel": dom,
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "occupancy": occupancy,
        "measured_in": f"a second pass of the same {args.steps} steps with one hipEventRecord after every kernel on the handle's stream ({1e3 * elapsed_profiled / args.steps:.1f} ms per step; the timed pass runs without them, and in parts on two streams where the batch is large enough: config.streams_per_gpu -- the profiled pass is one stream, whole batch per launch)",
        "avg_launch_ms": per_kernel[dom]["avg_launch_ms"],
        "bytes_per_unit": bytes_k[dom],
        "units_per_launch_avg": units / launches,
        "launches": launches,
        "all_kernels": per_kernel,
        "note": "units = running instances x (T-2) summed over the batched launches; iterations run inside the persistent tail kernel are excluded from units and time alike; k_eval is the pair of launches k_retract + k_evalb (rocprof lists them separately, profiles/*_kernel_stats.csv adds the pair)",
    }
    fk_achieved = nfk * BYTES_FKJAC / (fk_ms * 1e-3) / 1e9
    out = {
        "metric": "MPC solves/sec (KUKA 7-DoF, T=50)",
        "value": world * B * args.steps / elapsed,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"figure_eight_plan T=50 KUKA LWR end_effector_ball (BASELINE configs[1]) x {B} perturbed instances per GPU "
            f"(qc0 + U(-0.1,0.1)^7, seed=qc), solved to reduced-gradient tol {args.tol:g}, max_iter {args.max_iter}",
            "batch_per_gpu": B,
            "global_batch": world * B,
            "T": T,
            "parallelism": f"dp{world} (instances sharded, one RCCL broadcast of constants)",
            "rccl_world": rccl_world,
            **({"rccl_error": rccl_error} if rccl_error else {}),
            "per_rank": per_rank,
            "hessian": args.hessian,
            "streams_per_gpu": streams_used,
            "streams_note": "a batch at or above the handle's split_min is solved in `streams` contiguous parts, each on a HIP stream and host thread of its own "
                            "(csrc/oh_api.hip:solve_split): the latency-bound phases of one part overlap the bandwidth-bound launches of the other",
        },
        "roofline": roofline,
        "roofline_fk_jac": {
            "kernel": "k_fk_jac<SOA>",
            "bound": "hbm",
            "achieved": fk_achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": fk_achieved / HBM_PEAK_GBS,
            "units": nfk,
            "bytes_per_unit": BYTES_FKJAC,
            "avg_launch_ms": fk_ms,
            "reference_layout": {"kernel": "k_fk_jac<AoS> (q[n][ndof], pose[n][7], J[n][6][ndof], staged through LDS)", "avg_launch_ms": fk_ref_ms,
                                 "achieved": nfk * BYTES_FKJAC / (fk_ref_ms * 1e-3) / 1e9, "frac": nfk * BYTES_FKJAC / (fk_ref_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        },
        "quality": {
            "converged_frac": float(conv.mean()),
            "iters_p50": float(np.percentile(iters, 50)),
            "iters_p90": float(np.percentile(iters, 90)),
            "iters_max": int(iters.max()),
            "kkt_stationarity_max_converged": float(kkt[conv, 0].max()) if conv.any() else None,
            "feasibility_max": float(kkt[:, 1].max()),
            "f_mean": float(fvals.mean()),
            "oracle_sample": osample,
        },
        "device_ms_per_step": solve_ms_plain / args.steps,
        "specialized_kernels": {**spec_info, "note": "k_retract / k_evalb / k_tail compiled with hiprtc behind a constexpr copy of the handle's kinematic chain (oh_specialize; automatic at the first solve of >= 4096 instances, before the timed region)"},
        "latency_b1_ms": lat[1],
        "latency_b1024_ms": lat[min(1024, B)],
        "latency_note": None if args.timed_only else f"whole solve on the device, inputs resident, median of 7: B=1 ({lat['iters_1']:.0f} iterations), B={min(1024, B)} (mean {lat[f'iters_{min(1024, B)}']:.1f} iterations)",
        "kernel_ms_per_step": {"k_eval": tm["eval_ms"] / args.steps, "k_couple": tm["couple_ms"] / args.steps, "k_step": tm["step_ms"] / args.steps},
        "rejected_step_frac": tm["rejected_steps"] / max(1, tm["instance_launches"] + tm["tail_iterations"]),
        "tail_iteration_frac": tm["tail_iterations"] / max(1, tm["instance_launches"] + tm["tail_iterations"]),
        "compactions_per_step": tm["compactions"] / args.steps,
        "fused_coupling": zc,
    }
    def tol_block(tol, ms, it, st, note):
        return {"tol": tol, "value": B / (1e-3 * ms), "unit": "solves/s", "ms_per_step": ms, "iters_p50": float(np.percentile(it, 50)), "iters_p90": float(np.percentile(it, 90)),
                "iters_max": int(it.max()), "converged_frac": float((st == 0).mean()), "note": note}

    tol_key = lambda t: f"tol_{t:g}"
    out[tol_key(args.tol)] = tol_block(args.tol, solve_ms_plain / args.steps, iters, status, "the timed region of this line (device time of one step, HIP events around the solve)")
    if world == 1 and not args.timed_only:
        # the same K steps at the other of the two tolerances people quote for this path: 1e-8 = IPOPT's default `tol` = what the reference's configs run with
        # (no options passed: figure_eight_plan.py:111), 1e-6 = what rounds 1-5 of this repository quoted.  Option "tol" of the handle, nothing else changes.
        other = 1e-6 if args.tol != 1e-6 else 1e-8
        be.set_option("tol", other)
        be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
        ms_o = 0.0
        for _ in range(args.steps):
            be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
            ms_o += be.timing()["solve_ms"]
        f_o = d_f.download(np.float64, (B,))
        out[tol_key(other)] = tol_block(other, ms_o / args.steps, d_it.download(np.int32, (B,)), d_st.download(np.int32, (B,)),
                                        f"second pass of the same {args.steps} steps with oh_set_option(h, 'tol', {other:g}); device time")
        rel = np.abs(f_o - fvals) / np.abs(fvals)
        out[tol_key(other)]["objective_vs_headline_tol"] = {"max_rel_diff": float(rel.max()), "instances_beyond_1e-6": int((rel > 1e-6).sum()), "instances_beyond_1e-9": int((rel > 1e-9).sum())}
        be.set_option("tol", 0.0)
    if world == 1 and not args.timed_only:
        # the same batch with `batch_invariant` (every answer a function of the instance alone, bit for bit): what the option costs, and how many
        # answers of the default schedule it changes (not part of `value`)
        be.set_option("batch_invariant", 1)
        inv_ms = []
        for _ in range(3):
            be.solve_device(B, d_x0, d_p, d_x, d_f, d_k, d_it, d_st)
            inv_ms.append(be.timing()["solve_ms"])
        f_inv, st_inv = d_f.download(np.float64, (B,)), d_st.download(np.int32, (B,))
        t_inv = be.timing()
        be.set_option("batch_invariant", 0)
        out["batch_invariant"] = {
            "ms_per_step": float(np.median(inv_ms[1:])),
            "value": B / (1e-3 * float(np.median(inv_ms[1:]))),
            "unit": "solves/s",
            "cost_vs_default": float(np.median(inv_ms[1:])) / (solve_ms_plain / args.steps),
            "converged_frac": float((st_inv == 0).mean()),
            "same_optimum_as_default_frac": float((np.abs(f_inv - fvals) <= 1e-9 * np.abs(fvals)).mean()),
            "instances_with_another_optimum": int((np.abs(f_inv - fvals) > 1e-9 * np.abs(fvals)).sum()),
            "compactions": t_inv["compactions"],
            "iterations_launched": t_inv["iterations_launched"],
            "note": f"device time of one {B}-instance solve with oh_set_option(h, 'batch_invariant', 1): no restarts, no persistent kernel, survivors moved with everything they own (csrc/oh_api.hip:move_everything); bit-identity alone / in a batch / with and without compaction is asserted in tests/test_gpu_options.py",
        }
    out["pcie_inclusive"] = pcie
    if world == 1 and not args.no_configs and not args.timed_only:
        # the other BASELINE configs at their stated sizes (device ms, convergence, an oracle-graded sample each): tools/bench_configs.py
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs

        be.close()
        t0c = time.perf_counter()
        try:  # (the headline line is printed whatever happens in here)
            out["configs"] = bench_configs.run_configs(sample=0 if args.no_cpu_baseline else 8)
        except Exception as e:  # noqa: BLE001
            import traceback

            out["configs"] = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
        out["configs"]["seconds"] = time.perf_counter() - t0c
    if cpu is not None:
        f_port = cpu.pop("_f_port")
        n_p = len(f_port)
        relp = np.abs(fvals[:n_p] - f_port) / np.maximum(1.0, np.abs(f_port))
        out["quality"]["population_vs_host_port"] = {
            "instances": int(n_p), "misses_1e-6": int((relp > 1e-6).sum()), "misses_1e-9": int((relp > 1e-9).sum()), "max_rel_diff": float(relp.max()),
            "what": f"|f_gpu - f_port| / max(1, |f_port|) (BASELINE.md section 3: target <= 1e-6) over the first {n_p} instances of the timed batch, default options, both at tol {args.tol:g}: "
                    "the objectives of the timed steps against those the compiled host port of the state machine (oracle/cpu_port, the cpu_baseline leg) reached from the same seeds",
        }
        out["cpu_baseline"] = cpu
    if comm is not None:
        comm.barrier()
        comm.destroy()
    sys.stdout.flush()
    print(json.dumps(out), file=result_out, flush=True)


if __name__ == "__main__":
    main()
