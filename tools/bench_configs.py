"""Device-time measurements of the BASELINE configs other than the headline (bench.py measures configs[1]):
config 1 (IK, example.py), config 3 (point-mass MPC tick), config 4 (dual_arm.py as shipped and the synthetic T=100 +
limits + spheres variant), config 5 (torque MPC with RNEA equality rows, T=30, B=8192 and B=1024 = one GPU's share of 8192 over 8).  One JSON line per config on stdout; inputs are synthetic (SURVEY 8(d) seeds), timings are the
handle's HIP-event solve time with buffers already resident (oh_solve_device).

  python tools/bench_configs.py > profiles/r02_configs.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optas_amd import _lib  # noqa: E402
from optas_amd.backend import FigureEightBackend, IKBackend, PointMassBackend, TorqueBackend  # noqa: E402
from optas_amd.models import RobotModel  # noqa: E402

SEED = 20260927
F64_PEAK_TFLOPS = 78.6  # MI355X f64 vector peak = f64 MFMA peak (157.3 TF f32 / 2; the families below run 4x4 ... 7x7 blocks on the vector pipes)
FLOPS_FILE = os.path.join(ROOT, "profiles", "configs_flops.json")  # f64 flop per work unit of each family's kernels, from rocprofv3 --pmc passes (tools/gpu_configs_pmc.sh)


def pub(r: dict) -> dict:
    return {k: v for k, v in r.items() if not k.startswith("_")}


def usable_cores() -> int:
    sys.path.insert(0, ROOT)
    import bench

    return bench.usable_cores()


def cpu_leg(config: str, arrays: dict, f_gpu_head, seconds: float = 6.0):
    """The family's numpy port on the same instances, 1 process and all usable cores (tools/cpu_legs.py, a subprocess: no HIP runtime in it)."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, f"{config}.npz")
        np.savez(path, **arrays)
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_legs.py"), path, config, str(usable_cores()), str(seconds)], capture_output=True, text=True, timeout=600)
            leg = json.loads(cp.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}
    fp = np.array(leg.pop("f_first"))
    n = min(len(fp), len(f_gpu_head))
    leg["gpu_vs_port_objective_rel_max"] = float((np.abs(fp[:n] - f_gpu_head[:n]) / np.maximum(1e-12, np.abs(fp[:n]))).max()) if n else None
    leg["gpu_vs_port_instances"] = int(n)
    return leg


def flop_roofline(family: str, kernel_ms: dict, units: float, unit_name: str):
    """roofline object of a family whose kernels are f64-arithmetic / latency bound: achieved = (f64 flop per work unit of the dominant kernel group, counted once
    by the SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 counters: profiles/configs_flops.json) x (work units of THIS run) / (that group's HIP-event time in THIS run)."""
    dom = max(kernel_ms, key=kernel_ms.get)
    r = {"kernel": dom, "bound": "mfma", "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None,
         "peak_note": "f64: the MFMA f64 peak equals the vector f64 peak on MI355X (78.6 TFLOP/s); these kernels run their small dense blocks on the vector pipes",
         "kernel_ms": kernel_ms, "work_units": units, "work_unit": unit_name}
    try:
        fl = json.load(open(FLOPS_FILE))[family]
        per_unit = fl["flop_per_unit"][dom]
        r["flop_per_unit"] = per_unit
        r["flop_source"] = {"file": "profiles/configs_flops.json", "profile": fl.get("tag"), "counted_at_units": fl.get("units"), "note": "wave-level instruction counts x 64 lanes (2 flop per FMA): all lanes counted as active"}
        r["achieved"] = per_unit * units / (kernel_ms[dom] * 1e-3) / 1e12
        r["frac"] = r["achieved"] / F64_PEAK_TFLOPS
        r["all_kernels"] = {k: fl["flop_per_unit"].get(k, 0.0) * units / (v * 1e-3) / 1e12 for k, v in kernel_ms.items() if v > 0}
    except Exception as e:  # noqa: BLE001
        r["flop_source"] = f"unavailable ({type(e).__name__}: {e})"
    return r


def timed(be, x0, p, reps=3):
    B = x0.shape[0]
    bufs = [_lib.DeviceBuffer(a.nbytes) for a in (x0, p)]
    bufs[0].upload(x0)
    bufs[1].upload(p)
    d_x, d_f, d_k = _lib.DeviceBuffer(x0.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B)
    d_i, d_s = _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)
    ms = []
    host = not hasattr(be, "solve_device")  # (a problem solved over fewer variables than it has: the host entry point puts the eliminated ones back)
    for _ in range(reps + 1):
        if host:
            rh = be.solve(x0, p)
        else:
            be.solve_device(B, bufs[0], bufs[1], d_x, d_f, d_k, d_i, d_s)
        ms.append(be.solve_ms() if hasattr(be, "solve_ms") else be.timing()["solve_ms"])
    if host:
        it, st, kk = rh.iters, rh.status, rh.kkt
        d_x.upload(rh.x)
        d_f.upload(rh.f)
    else:
        it, st, kk = d_i.download(np.int32, (B,)), d_s.download(np.int32, (B,)), d_k.download(np.float64, (B, 3))
    for b in bufs + [d_x, d_f, d_k, d_i, d_s]:
        b.free()
    ok = st == 0
    return {"ms": float(np.median(ms[1:])), "converged_frac": float(ok.mean()), "iters_p50": float(np.median(it)), "iters_max": int(it.max()),
            "stationarity_max": float(kk[ok, 0].max()), "feasibility_max": float(kk[ok, 1].max())}


def oracle_grade(kind, **kw):
    """A few instances of a config's batch graded by oracle/ (never by the library): reference-form KKT residuals on the literal layout
    (oracle/solvers.py:kkt_reference_form) where the literal NLP is small enough to grade in a second, feasibility of every inequality row and
    the recomputed objective otherwise."""
    from oracle.robot import OracleRobot
    from oracle.solvers import kkt_reference_form

    R = os.path.join(ROOT, "optas_amd", "robots")
    if kind == "ik":
        from oracle.problems import IKExampleNLP

        nlp = IKExampleNLP(OracleRobot(os.path.join(R, "kuka_lwr.kin.json")))
        ks = [kkt_reference_form(nlp, x, p, active_tol=1e-7) for x, p in zip(kw["x"], kw["p"])]
    elif kind == "pm":
        from oracle.problems import PointMassMPCNLP

        nlp = PointMassMPCNLP()
        ks = [kkt_reference_form(nlp, x, p, active_tol=1e-3) for x, p in zip(kw["x"], kw["p"])]
        return {"instances": len(ks), "stationarity_max": max(k["stationarity"] for k in ks), "feasibility_max": max(k["feasibility"] for k in ks),
                "objective_recomputed_max_abs_diff": float(max(abs(nlp.f(x, p) - f) for x, p, f in zip(kw["x"], kw["p"], kw["f"]))),
                "by": "oracle/solvers.py:kkt_reference_form on oracle/problems.py:PointMassMPCNLP (literal 264-row v)"}
    elif kind == "torque":
        from oracle.problems import TorqueMPCNLP
        from oracle.torque import TorqueProblem

        prob = TorqueProblem(OracleRobot(os.path.join(R, "med7.kin.json")), "lbr_link_ee", T=kw["T"], dt=0.1, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lim=kw["lim"])
        nlp = TorqueMPCNLP(prob)
        # an interior-point answer is graded with the multipliers it came with (lam_i = mu_b / s_i per knot [lo; up] -> the row order of k)
        lks = [np.concatenate([l[:, :7].reshape(-1), l[:, 7:].reshape(-1)]) for l in kw["lam"]]
        ks = [kkt_reference_form(nlp, x, p, lam_kg=lk) for x, p, lk in zip(kw["x"], kw["p"], lks)]
        return {"instances": len(ks), "stationarity_max": max(k["stationarity"] for k in ks), "feasibility_max": max(k["feasibility"] for k in ks),
                "complementarity_max": max(k["complementarity"] for k in ks), "inequality_rows_min": float(min(nlp.k(x, p).min() for x, p in zip(kw["x"], kw["p"]))),
                "linear_rows_max": float(max(np.abs(nlp.a(x, p)).max() for x, p in zip(kw["x"], kw["p"]))),
                "dynamics_rows_max": float(max(np.abs(nlp.h(x, p)).max() for x, p in zip(kw["x"], kw["p"]))),
                "objective_recomputed_max_rel_diff": float(max(abs(nlp.f(x, p) - f) / abs(f) for x, p, f in zip(kw["x"], kw["p"], kw["f"]))),
                "by": "oracle/solvers.py:kkt_reference_form on oracle/problems.py:TorqueMPCNLP (literal 840-variable / 1680-row layout, RNEA rows by the literal recursion), with the multipliers returned by oh_get_multipliers"}
    elif kind == "fig8_vel":
        from oracle.problems import LimitedFigureEightNLP

        orc = OracleRobot(os.path.join(R, "kuka_lwr.kin.json"))
        vl = np.asarray(orc.velocity_actuated_joint_limits)
        nlp = LimitedFigureEightNLP(orc, "end_effector_ball", vlo=-vl, vup=vl, T=50)
        ks = [kkt_reference_form(nlp, x, p, active_tol=1e-7) for x, p in zip(kw["x"], kw["p"])]
        return {"instances": len(ks), "stationarity_max": max(k["stationarity"] for k in ks), "feasibility_max": max(k["feasibility"] for k in ks),
                "complementarity_max": max(k["complementarity"] for k in ks),
                "velocity_rows_min": float(min(nlp.k(x, p).min() for x, p in zip(kw["x"], kw["p"]))),
                "objective_recomputed_max_rel_diff": float(max(abs(nlp.f(x, p) - f) / abs(f) for x, p, f in zip(kw["x"], kw["p"], kw["f"]))),
                "by": "oracle/solvers.py:kkt_reference_form on oracle/problems.py:LimitedFigureEightNLP (literal layout, 686 velocity rows)"}
    elif kind == "guarded_arm":
        from oracle.guarded import Guards, guard_values
        from oracle.structured import FoldedChain

        rob = OracleRobot(os.path.join(R, "kuka_lwr.kin.json"), name="kukal")
        rob.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        ch = FoldedChain(rob, "end_effector_ball")
        T = kw["T"]
        worst, fdiff = 0.0, 0.0
        for x, p, f in zip(kw["x"], kw["p"], kw["f"]):
            Q = x[: 7 * T].reshape(T, 7)
            dQ = x[7 * T :].reshape(T - 1, 7)
            G = Guards(lo=rob.lower_actuated_joint_limits, up=rob.upper_actuated_joint_limits, links=kw["links"], link_radii=p[7:11], obs_pos=p[11:].reshape(6, 4)[:, :3],
                       obs_radii=p[11:].reshape(6, 4)[:, 3])
            worst = max(worst, float(-min(0.0, guard_values(ch, Q, G)[0].min())))  # every knot, the pinned one included
            e = ch.fk(Q)[0]
            path = ch.fk(p[None, :7])[0][0][None] + kw["offsets"]
            fdiff = max(fdiff, abs(float(np.sum((e - path) ** 2) + 0.01 * np.sum(dQ**2)) - f))
        # reference-form KKT on the literal layout (round 4): two arms at a time as one dual-arm instance of oracle/problems.py:GuardedDualArmNLP
        # (both slots hold the arm of this batch: base (0, -0.25, 0), the left arm's path).  An arm whose pinned initial configuration breaks a
        # clearance poses an infeasible NLP (rows of knot 0: negative constants) and comes back OH_STATUS_INFEASIBLE; the generator draws none.
        from oracle.problems import GuardedDualArmNLP

        rob2 = OracleRobot(os.path.join(R, "kuka_lwr.kin.json"), name="kukar")
        rob2.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        nlp = GuardedDualArmNLP(rob, rob2, kw["links"], 6, T=T)
        nlp.offsets = {"l": kw["offsets"].T, "r": kw["offsets"].T}
        G0 = lambda p: Guards(lo=None, up=None, links=kw["links"], link_radii=p[7:11], obs_pos=p[11:].reshape(6, 4)[:, :3], obs_radii=p[11:].reshape(6, 4)[:, 3])
        feas0 = [i for i, p in enumerate(kw["p"]) if guard_values(ch, p[None, :7], G0(p))[0].min() >= 0.0]
        assert len(feas0) == len(kw["p"]), "the instance generator draws by rejection: every sampled arm must be feasible as posed"
        ks = []
        for a, b in zip(feas0[0::2], feas0[1::2]):
            x2 = np.concatenate([kw["x"][a], kw["x"][b]])
            p2 = np.concatenate([kw["p"][a][:7], kw["p"][b][:7], kw["p"][a][7:], kw["p"][b][7:]])
            assert abs(nlp.f(x2, p2) - (kw["f"][a] + kw["f"][b])) <= 1e-9
            ks.append(kkt_reference_form(nlp, x2, p2, active_tol=1e-6))
        return {"instances": len(kw["x"]), "inequality_rows_min_violation": worst, "objective_recomputed_max_abs_diff": fdiff,
                "infeasible_as_posed": len(kw["x"]) - len(feas0), "kkt_graded_arms": 2 * len(ks),
                "stationarity_max": max((k["stationarity"] for k in ks), default=None), "complementarity_max": max((k["complementarity"] for k in ks), default=None),
                "by": "oracle/guarded.py:guard_values (2 x 7 limit rows + 4 x 6 sphere rows per knot), the tracking cost recomputed with oracle/structured.py:FoldedChain, and "
                      "oracle/solvers.py:kkt_reference_form on oracle/problems.py:GuardedDualArmNLP (literal layout, two arms per instance) for the sampled arms (all feasible as posed: drawn by rejection)"}
    return {"instances": len(ks), "stationarity_max": max(k["stationarity"] for k in ks), "feasibility_max": max(k["feasibility"] for k in ks),
            "complementarity_max": max(k["complementarity"] for k in ks), "by": "oracle/solvers.py:kkt_reference_form on the literal NLP of oracle/problems.py"}


PROBE = None  # tools/gpu_configs_pmc.sh: {"units": work units of every solve run so far} -- the denominator of the flop counters rocprofv3 collects over the same process


def _probe_units(be, it):
    if PROBE is None:
        return
    tm = be.timing() if hasattr(be, "timing") else {}
    w = tm.get("work_instances", tm.get("instance_launches", 0.0))
    PROBE["units"] += float(w) if w else float(np.asarray(it).sum())
    PROBE["solves"] += 1


def timed_with_results(be, x0, p, reps=3, sample=0, seed=0, profile=False):  # (median of three timed solves after one warm-up: with two, one hiccup of the box moves the figure)
    """timed() plus a sample of (x, p, f) rows downloaded from the device for the oracle."""
    B = x0.shape[0]
    bufs = [_lib.DeviceBuffer(a.nbytes) for a in (x0, p)]
    bufs[0].upload(x0)
    bufs[1].upload(p)
    d_x, d_f, d_k = _lib.DeviceBuffer(x0.nbytes), _lib.DeviceBuffer(8 * B), _lib.DeviceBuffer(24 * B)
    d_i, d_s = _lib.DeviceBuffer(4 * B), _lib.DeviceBuffer(4 * B)
    ms = []
    host = not hasattr(be, "solve_device")  # (a problem solved over fewer variables than it has: the host entry point puts the eliminated ones back)
    if PROBE is not None and profile and hasattr(be, "set_profiling"):
        be.set_profiling(True)  # the counters are collected on the launches the roofline pass times: one stream, an event after every kernel
    for _ in range(reps + 1):
        if host:
            rh = be.solve(x0, p)
        else:
            be.solve_device(B, bufs[0], bufs[1], d_x, d_f, d_k, d_i, d_s)
        ms.append(be.solve_ms() if hasattr(be, "solve_ms") else be.timing()["solve_ms"])
        if PROBE is not None:
            _probe_units(be, rh.iters if host else d_i.download(np.int32, (B,)))
    if PROBE is not None and profile and hasattr(be, "set_profiling"):
        be.set_profiling(False)
    if host:
        it, st, kk = rh.iters, rh.status, rh.kkt
        d_x.upload(rh.x)
        d_f.upload(rh.f)
    else:
        it, st, kk = d_i.download(np.int32, (B,)), d_s.download(np.int32, (B,)), d_k.download(np.float64, (B, 3))
    ok = st == 0
    r = {"device_ms": float(np.median(ms[1:])), "converged_frac": float(ok.mean()), "iters_p50": float(np.median(it)), "iters_p90": float(np.percentile(it, 90)),
         "iters_max": int(it.max()), "stationarity_max": float(kk[ok, 0].max()), "feasibility_max": float(kk[ok, 1].max())}
    F = d_f.download(np.float64, (B,))
    r["_f_head"], r["_iters_sum"] = F[:8].copy(), float(it.sum())
    if profile and not host and hasattr(be, "set_profiling"):
        # one more solve with a HIP event after every kernel of the handle's stream (one stream, no split): the per-kernel times of the roofline object
        be.set_profiling(True)
        be.solve_device(B, bufs[0], bufs[1], d_x, d_f, d_k, d_i, d_s)
        tp = be.timing()
        if PROBE is not None:
            _probe_units(be, None)
        be.set_profiling(False)
        r["_profiled"] = {"eval_ms": tp.get("eval_ms", 0.0), "step_ms": tp.get("step_ms", 0.0), "solve_ms": tp["solve_ms"],
                          "work": float(tp.get("work_instances", tp.get("instance_launches", 0.0)))}
    smp = None
    if sample:
        X = d_x.download(np.float64, x0.shape)
        idx = np.sort(np.random.default_rng(seed).choice(np.flatnonzero(ok), min(sample, int(ok.sum())), replace=False))
        smp = {"x": X[idx], "p": p[idx], "f": F[idx], "idx": idx}
    for b in bufs + [d_x, d_f, d_k, d_i, d_s]:
        b.free()
    return r, smp


def run_configs(sample=8, torque_batches=(8192, 1024), only=None, cpu=None, ik_batch=65536, pm_batch=4096, config4_cases=((256, 0.15), (1024, 0.15), (1024, 0.1))):
    """BASELINE configs 1, 3, 4, 5 at their stated sizes: device time of one batched solve (HIP events, inputs resident), convergence, and an
    oracle-graded sample of each -- what bench.py prints as its `configs` block."""
    rng = np.random.default_rng(SEED)
    out = {}
    kuka = RobotModel.builtin("kuka_lwr")
    cpu = (sample > 0) if cpu is None else cpu  # CPU legs (numpy ports on the host cores) go with the oracle-graded samples: both are off under --no-cpu-baseline
    if only == "torque":
        rng = np.random.default_rng(SEED + 5)
        return _torque(out, rng, sample, torque_batches, cpu)
    if only == "config4":
        return _config4(out, np.random.default_rng(SEED + 4), sample, cpu, config4_cases)
    if only == "pm":
        return _pm(out, rng, sample, cpu, pm_batch)
    # config 1
    B = ik_batch
    be = IKBackend(kuka.kinematic_chain("end_effector_ball"), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits, max_iter=300)
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits,
                                                                             kuka.upper_actuated_joint_limits).T)).T
    r, smp = timed_with_results(be, np.ascontiguousarray(qn), np.ascontiguousarray(np.concatenate([qn, pg], 1)), sample=sample, seed=1)
    out["config1_ik"] = {"what": f"example.py IK (KUKA LWR, joint limits, position goal), B = {B}", "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, **pub(r),
                         "oracle_sample": oracle_grade("ik", **smp) if smp else None,
                         "roofline": flop_roofline("ik", {"k_ik (the whole solve: one launch)": r["device_ms"]}, r["_iters_sum"], "instance-step")}
    if cpu:
        nl = min(B, 2048)
        out["config1_ik"]["cpu_baseline"] = cpu_leg("ik", dict(n=nl, qn=qn[:nl], pg=pg[:nl], lo=np.asarray(kuka.lower_actuated_joint_limits, float), up=np.asarray(kuka.upper_actuated_joint_limits, float)), r["_f_head"])
    be.close()
    if only == "ik":
        return out
    _pm(out, rng, sample, cpu, pm_batch)
    _velocity_limited(out, sample)
    _config4(out, rng, sample, cpu, config4_cases)
    _planner_tape(out, sample)
    return _torque(out, rng, sample, torque_batches, cpu)


def pm_family(be, B):
    """The point-mass solve is one wavefront per plant (lane = knot, k_pm_solve_wave) up to pm_wave_max plants and one thread per plant beyond (k_pm_solve):
    two kernels, two flop-per-step constants in profiles/configs_flops.json."""
    return "pm" if B <= int(be.get_option("pm_wave_max")) else "pm_thread"


def _pm(out, rng, sample, cpu=False, B=4096):
    # config 3: tick and closed loop
    from examples.point_mass_mpc import obstacle_and_goal

    be = PointMassBackend()
    P = []
    obs, _ = obstacle_and_goal(2.0, np.zeros(2))
    if B <= 65536:
        while len(P) < B:
            c = rng.uniform(-1.2, 1.2, 2)
            if np.linalg.norm(c - obs[:, 0]) <= 0.35:
                continue
            goal = np.stack([np.clip(c[j] + (1 - c[j]) * np.arange(20) / 19.0, -1.5, 1.5) for j in range(2)])
            P.append(np.concatenate([c, np.zeros(2), goal.T.reshape(-1), obs.T.reshape(-1)]))
        P = np.array(P)
    else:  # (batch sweeps: the same distribution drawn in bulk)
        c = rng.uniform(-1.2, 1.2, (2 * B, 2))
        c = c[np.linalg.norm(c - obs[:, 0][None], axis=1) > 0.35][:B]
        goal = np.clip(c[:, None, :] + (1 - c[:, None, :]) * (np.arange(20) / 19.0)[None, :, None], -1.5, 1.5)  # [B][20][2]
        P = np.ascontiguousarray(np.concatenate([c, np.zeros((B, 2)), goal.reshape(B, -1), np.tile(obs.T.reshape(-1), (B, 1))], 1))
    r, smp = timed_with_results(be, np.zeros((B, 80)), P, sample=sample, seed=3)
    n_ticks, adv, T = 50, 2, 20
    tab = np.array([[0.15 * np.sin((2.0 + 0.05 * j) * np.pi - np.pi), 0.15 * np.cos((2.0 + 0.05 * j) * np.pi - np.pi) + 0.15] for j in range(n_ticks * adv + T)])
    if PROBE is not None or B != 4096:  # (counter passes: only the launches whose work units are counted; batch sweeps: the tick alone)
        out["config3_point_mass"] = {"what": f"point_mass_mpc.py tick, B = {B}", "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, **pub(r),
                                     "roofline": flop_roofline(pm_family(be, B), {"k_pm (the whole solve: one launch)": r["device_ms"]}, r["_iters_sum"], "instance-step")}
        be.close()
        return out
    be.rollout(P[:, :4], tab, 2)
    _, _, _, stt = be.rollout(P[:, :4], tab, n_ticks, adv)
    out["config3_point_mass"] = {"what": f"point_mass_mpc.py tick (T=20, box limits, moving obstacle), B = {B} initial states", "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, **pub(r),
                                 "oracle_sample": oracle_grade("pm", **smp) if smp else None,
                                 "closed_loop": {"ticks": n_ticks, "device_ms": be.solve_ms(), "ticks_per_s": B * n_ticks / be.solve_ms() * 1e3, "converged_frac": float((stt == 0).mean())},
                                 "roofline": flop_roofline(pm_family(be, B), {"k_pm (the whole solve: one launch)": r["device_ms"]}, r["_iters_sum"], "instance-step")}
    if cpu:
        nl = min(B, 2048)
        out["config3_point_mass"]["cpu_baseline"] = cpu_leg("pm", dict(n=nl, P=P[:nl]), r["_f_head"])
    be.close()
    return out


def _velocity_limited(out, sample):
    # config 2 with enforce_model_limits(name, time_deriv=1) (the shipped script's optimum breaks the LWR's speed limit on joint 0)
    from examples.figure_eight_plan import setup_solver

    B = 65536
    qcs = np.deg2rad([0, 30, 0, -90, 0, -30, 0])[None] + np.random.default_rng(SEED + 2).uniform(-0.1, 0.1, (B, 7))
    kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
    x0 = np.zeros((B, solver.opt.nx))
    x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
    r, smp = timed_with_results(solver.backend, x0, np.ascontiguousarray(qcs), sample=min(sample, 4), seed=2)
    out["config2_velocity_limited"] = {"what": "figure_eight_plan.py T=50 + joint-velocity limits (686 inequality rows), B = 65536; persistent kernel k_tail_vel", "batch": B,
                                       "solves_per_s": B / r["device_ms"] * 1e3, **pub(r), "oracle_sample": oracle_grade("fig8_vel", **smp) if smp else None}
    solver.backend.close()


def _planner_tape(out, sample):
    # SURVEY 8(f)1, the generic route: example/simple_joint_space_planner.py (280 variables, 40 + 154 rows) matches no hand-written family and runs on the
    # tape family -- one block of four wavefronts per instance over the dependency levels of the tape (csrc/oh_tape_wave.hip)
    from examples.simple_joint_space_planner import setup_solver

    g = np.load(os.path.join(ROOT, "tests", "golden", "planner_golden.npz"))
    robot, solver = setup_solver(solver_options={"max_iter": 400000})
    be = solver.backend
    P, nb = g["p"], len(g["p"])
    x0 = np.zeros((nb, solver.opt.nx))
    x0[:, :140] = np.tile(g["q0"].reshape(-1, 1), (1, 20)).reshape(-1)[None, :]
    be.solve(x0, P)
    r4 = be.solve(x0, P)
    ms4 = float(be.solve_ms())
    rng = np.random.default_rng(SEED + 6)
    B = 256
    idx = np.arange(B) % nb
    Pn = P[idx].copy()
    Pn[:, :14] += rng.uniform(-0.05, 0.05, (B, 14))
    Pn[:, 14:17] += rng.uniform(-0.02, 0.02, (B, 3))
    r, smp = timed_with_results(be, np.ascontiguousarray(x0[idx]), np.ascontiguousarray(Pn), reps=3, sample=min(sample, 4), seed=6)
    grade = None
    if smp:
        from oracle.problems import JointSpacePlannerNLP
        from oracle.robot import OracleRobot

        nlp = JointSpacePlannerNLP(OracleRobot(os.path.join(ROOT, "optas_amd", "robots", "med7.kin.json")))
        grade = {"n": int(len(smp["f"])), "f_recomputed_max_abs_diff": float(max(abs(nlp.f(x, p) - f) for x, p, f in zip(smp["x"], smp["p"], smp["f"]))),
                 "equality_rows_max": float(max(max(np.abs(nlp.a(x, p)).max(), np.abs(nlp.h(x, p)).max()) for x, p in zip(smp["x"], smp["p"]))),
                 "inequality_rows_min": float(min(nlp.g(x, p).min() for x, p in zip(smp["x"], smp["p"]))),
                 "by": "oracle/problems.py:JointSpacePlannerNLP (literal layout: 40 inequality, 147 + 7 equality rows)"}
    out["planner_tape"] = {"what": "simple_joint_space_planner.py (280 variables, 154 equality rows of which the host eliminates the 147 affine ones: 133 variables on the device) on the generic tape family, one block of wavefronts per instance; B = 256 perturbed problems",
                           "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, **pub(r),
                           "path": {k: be.flag(k) for k in ("tape_wave", "tape_levels", "tape_passes")},
                           "golden_instances": {"n": nb, "device_ms": ms4, "evaluations": [int(v) for v in r4.iters], "converged": bool((np.asarray(r4.status) == 0).all()),
                                                "f_rel_diff_to_interior_point_golden": [float(abs(a - b) / b) for a, b in zip(r4.f, g["f"])]},
                           "oracle_sample": grade}
    be.close()


def _config4(out, rng, sample, cpu=False, cases=((256, 0.15), (1024, 0.15), (1024, 0.1))):
    # config 4 synthetic: T = 100, limits + 4 x 6 sphere rows per knot, link radius 0.15 as SURVEY 8(d) states; arms are independent instances
    from examples.dual_arm import SPHERE_LINKS, draw_feasible_configurations, path_offsets

    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    T = 100
    offs = path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3])
    for B, radius in cases:
        arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
        arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        g = _lib.oh_guards()
        g.limits = 1
        for j in range(7):
            g.q_lo[j], g.q_up[j] = arm.lower_actuated_joint_limits[j], arm.upper_actuated_joint_limits[j]
        g.n_links, g.n_obstacles = 4, 6
        for l, (k, off) in enumerate(arm.link_attachments("end_effector_ball", SPHERE_LINKS)):
            g.link_joint[l] = k
            for i in range(3):
                g.link_offset[l][i] = off[i]
        be = FigureEightBackend(arm.kinematic_chain("end_effector_ball"), T, 10.0 / (T - 1), offs.T, w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False,
                                path_in_frame=False, guards=g)
        # perturbed initial configurations by rejection: every arm is feasible as posed (q_0 = qc is pinned, so the clearances of knot 0 are constants;
        # round-4 verdict, Weak 3 -- the library reports the other kind as OH_STATUS_INFEASIBLE)
        qc = draw_feasible_configurations(rng, B, arm, link_radius=radius)
        obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
        p = np.ascontiguousarray(np.concatenate([qc, np.full((B, 4), radius), np.tile(obs_row, (B, 1))], 1))
        x0 = np.ascontiguousarray(np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1))
        r, smp = timed_with_results(be, x0, p, sample=max(sample, 16) if sample else 0, seed=4, profile=True)
        key4 = f"config4_arms{B}_r{radius:g}"
        out[key4] = {"what": f"dual_arm.py synthetic: T=100, joint limits + 4 x 6 sphere clearances (link radius {radius:g}), {B} arms "
                                                        "(a dual-arm instance = two of them)", "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, **pub(r),
                                               "oracle_sample": oracle_grade("guarded_arm", T=T, links=SPHERE_LINKS, offsets=offs.T, **smp) if smp else None}
        pr = r.get("_profiled")
        if pr:
            out[key4]["roofline"] = flop_roofline("guarded", {"k_eval_guarded": pr["eval_ms"], "k_step_free*": pr["step_ms"]}, pr["work"], "instance-launch (T - 1 knots)")
            out[key4]["roofline"]["profiled_solve_ms"] = pr["solve_ms"]
        if cpu and B == 1024:
            nl = min(B, 512)
            out[key4]["cpu_baseline"] = cpu_leg("guarded", dict(n=nl, p=p[:nl], T=T, dt=10.0 / (T - 1), offsets=offs.T, links=np.array(SPHERE_LINKS)), r["_f_head"])
        be.close()
    return out


def _torque(out, rng, sample, torque_batches, cpu=False):
    # config 5
    med7 = RobotModel.builtin("med7")
    link, T, dt = "lbr_link_ee", 30, 0.1
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    ts = np.arange(T) * dt
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
    for B in tuple(torque_batches) + (1,):
        qc = qn + (rng.uniform(-0.1, 0.1, (B, 7)) if B > 1 else 0.0)
        qc = np.atleast_2d(qc)
        pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
        x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
        Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                       np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                       np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
        goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
        p = np.ascontiguousarray(np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1))
        x0 = np.zeros((B, 4 * 7 * T))
        x0[:, : 7 * T] = np.tile(qc, (1, T))
        be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
        r, smp = timed_with_results(be, x0, p, sample=((min(sample, 4) if B > 1 else 1) if sample else 0), seed=5, reps=3 if B > 1 else 5, profile=True)  # (median of three: a latency-bound solve of 80 launches shows any hiccup of the host loop)
        tm = be.timing()
        if smp:
            smp["lam"] = be.multipliers(B)[smp["idx"]]
        out[f"config5_torque_b{B}"] = {"what": f"torque MPC, RNEA dynamics equality rows + effort limits 58 N m (med7, T=30), primal-dual interior point, B = {B}" + (" (the nominal instance)" if B == 1 else ""),
                                       "batch": B, "solves_per_s": B / r["device_ms"] * 1e3, "iterations_launched": tm["iterations_launched"], **pub(r),
                                       "oracle_sample": oracle_grade("torque", T=T, lim=58.0, **smp) if smp else None}
        pr = r.get("_profiled")
        if pr:
            out[f"config5_torque_b{B}"]["roofline"] = flop_roofline("torque", {"k_tq_eval3+k_tq_curv": pr["eval_ms"], "k_tq_step": pr["step_ms"]}, pr["work"], "instance-iteration (T = 30 knots)")
            out[f"config5_torque_b{B}"]["roofline"]["profiled_solve_ms"] = pr["solve_ms"]
        if cpu and B == max(torque_batches):
            nl = min(B, 256)
            out[f"config5_torque_b{B}"]["cpu_baseline"] = cpu_leg("torque", dict(n=nl, qc=qc[:nl], goal=goal[:nl], T=T, dt=dt, lim=58.0), r["_f_head"], seconds=9.0)
        if B == max(torque_batches) and PROBE is None:
            # the MPC steady state (round 5, oh_tq_rollout): the same plants in closed loop, warm-started ticks resident on the device.  The goal table
            # continues the figure of eight; the plant follows each plan for one knot.  Graded: a warm tick ends where the cold solve from the same state does.
            n_ticks = 50
            ts2 = np.arange(n_ticks + T) * dt
            loc2 = np.stack([0.2 * np.sin(ts2 * np.pi * 0.5), 0.1 * np.sin(ts2 * np.pi), np.zeros(n_ticks + T)])
            table = np.ascontiguousarray(pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc2))
            st0 = np.concatenate([qc, np.zeros((B, 7))], 1)
            be.rollout(st0[:64], np.ascontiguousarray(table[:64, : 2 + T]), 2)  # warm-up of the code path
            states, tau0, fr, itr, stt = be.rollout(st0, table, n_ticks)
            ms = be.timing()["solve_ms"]
            idx = np.sort(np.random.default_rng(55).choice(B, 32, replace=False))
            k = n_ticks // 2
            pk = np.ascontiguousarray(np.concatenate([states[k, idx], table[idx, k : k + T].reshape(len(idx), -1)], 1))
            xk = np.zeros((len(idx), 4 * 7 * T))
            xk[:, 2 * 7 * T : 2 * 7 * T + 7] = -states[k, idx, 7:] / dt  # cold seed: brake to rest, hold still
            cold = be.solve(xk, pk)
            okc = _lib.status_ok(cold.status)
            out["config5_torque_closed_loop"] = {
                "what": f"torque MPC in closed loop (oh_tq_rollout): {B} plants x {n_ticks} ticks, seed = previous plan shifted one knot, barrier parameter of warm ticks 1e-6, plant = the plan's next state",
                "batch": B, "ticks": n_ticks, "device_ms": ms, "ticks_per_s": B * n_ticks / ms * 1e3, "ms_per_tick": ms / n_ticks,
                "converged_frac": float(_lib.status_ok(stt).mean()), "steps_cold_tick_p50": float(np.median(itr[0])), "steps_warm_tick_p50": float(np.median(itr[1:])),
                "steps_warm_tick_p90": float(np.percentile(itr[1:], 90)), "steps_warm_tick_max": int(itr[1:].max()),
                "warm_vs_cold_objective_rel_max": float((np.abs(cold.f - fr[k, idx]) / np.abs(cold.f))[okc].max()), "cold_steps_same_states_p50": float(np.median(cold.iters[okc]))}
        be.close()
    return out


def probe_main(family):
    """One family's GPU leg (no oracle, no CPU leg), run under rocprofv3 --pmc by tools/gpu_configs_pmc.sh: prints the work units of all its solves."""
    global PROBE
    PROBE = {"units": 0.0, "solves": 0}
    run_configs(sample=0, cpu=False, only={"ik": "ik", "pm": "pm", "pm_thread": "pm", "guarded": "config4", "torque": "torque"}[family], torque_batches=(8192,),
                pm_batch=65536 if family == "pm_thread" else 4096)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "cfgpmc"), exist_ok=True)
    json.dump({"family": family, **PROBE}, open(os.path.join(ROOT, "gpurun_out", "cfgpmc", f"{family}_units.json"), "w"))
    print(json.dumps(PROBE))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--probe":
        return probe_main(sys.argv[2])
    rng = np.random.default_rng(SEED)
    out = []
    kuka = RobotModel.builtin("kuka_lwr")
    # ---- config 1: IK ----
    B = 65536
    be = IKBackend(kuka.kinematic_chain("end_effector_ball"), kuka.lower_actuated_joint_limits, kuka.upper_actuated_joint_limits, max_iter=300)
    qn = np.deg2rad([0, 45, 0, -90, 0, -45, 0]) + rng.uniform(-0.3, 0.3, (B, 7))
    pg = np.asarray(kuka.get_global_link_position("end_effector_ball", np.clip(qn + rng.uniform(-0.5, 0.5, (B, 7)), kuka.lower_actuated_joint_limits,
                                                                             kuka.upper_actuated_joint_limits).T)).T
    r = timed(be, np.ascontiguousarray(qn), np.ascontiguousarray(np.concatenate([qn, pg], 1)))
    out.append({"config": "1 example.py IK (KUKA LWR, joint limits, position goal)", "batch": B, "solves_per_s": B / r["ms"] * 1e3, **r})
    # ---- config 1 again through the generic tape family (interpreter + BFGS instead of the hand-written kernel + Newton) ----
    sys.path.insert(0, ROOT)
    from examples.example import setup_solver as ik_setup
    from optas_amd.backend import TapeBackend
    from optas_amd.tape import compile_problem

    tp = compile_problem(ik_setup(build_only=True)[1])
    r = timed(TapeBackend(tp), np.ascontiguousarray(qn), np.ascontiguousarray(np.concatenate([qn, pg], 1)))
    out.append({"config": "1 example.py IK lowered to the generic tape family (OH_PROBLEM_TAPE)", "batch": B, "tape_instructions": len(tp.op),
                "solves_per_s": B / r["ms"] * 1e3, **r})
    # ---- config 3: point-mass MPC tick ----
    from examples.point_mass_mpc import obstacle_and_goal

    B = 4096
    be = PointMassBackend()
    P = []
    obs, _ = obstacle_and_goal(2.0, np.zeros(2))
    while len(P) < B:
        c = rng.uniform(-1.2, 1.2, 2)
        if np.linalg.norm(c - obs[:, 0]) <= 0.35:
            continue
        goal = np.stack([np.clip(c[j] + (1 - c[j]) * np.arange(20) / 19.0, -1.5, 1.5) for j in range(2)])
        P.append(np.concatenate([c, np.zeros(2), goal.T.reshape(-1), obs.T.reshape(-1)]))
    r = timed(be, np.zeros((B, 80)), np.array(P))
    out.append({"config": "3 point_mass_mpc.py tick (T=20, box limits, moving obstacle)", "batch": B, "solves_per_s": B / r["ms"] * 1e3, **r})
    # ---- config 3 closed loop: 50 ticks on the device vs the same loop with one oh_solve per tick from the host ----
    import time

    n_ticks, adv, T = 50, 2, 20
    tab = np.array([[0.15 * np.sin((2.0 + 0.05 * j) * np.pi - np.pi), 0.15 * np.cos((2.0 + 0.05 * j) * np.pi - np.pi) + 0.15] for j in range(n_ticks * adv + T)])
    st0 = np.array([p[:4] for p in P])
    be.rollout(st0, tab, 2)
    t0 = time.perf_counter()
    states, f, it, stt = be.rollout(st0, tab, n_ticks, adv)
    wall = time.perf_counter() - t0
    dev_ms = be.solve_ms()
    t0 = time.perf_counter()
    st, x0 = st0.copy(), np.zeros((B, 80))
    for k in range(n_ticks):
        goal = st[:, None, :2] + 0.032 * np.arange(T)[None, :, None]
        pk = np.concatenate([st, goal.reshape(B, -1), np.tile(tab[k * adv : k * adv + T].reshape(-1), (B, 1))], 1)
        r = be.solve(x0, pk)
        x0 = r.x
        st = np.concatenate([r.x[:, 2 * adv : 2 * adv + 2], r.x[:, 2 * T + 2 * adv : 2 * T + 2 * adv + 2]], 1)
    host_wall = time.perf_counter() - t0
    out.append({"config": "3 closed loop (oh_pm_rollout): plants x ticks, warm-started, device resident", "batch": B, "ticks": n_ticks,
                "ticks_per_s_device": B * n_ticks / dev_ms * 1e3, "device_ms": dev_ms, "wall_ms": wall * 1e3, "host_driven_loop_wall_ms": host_wall * 1e3,
                "converged_frac": float((stt == 0).mean()), "iters_mean": float(it.mean()), "max_state_diff_vs_host_loop": float(np.abs(st - states[-1]).max())})
    # ---- config 4 as shipped and synthetic ----
    from examples.dual_arm import SPHERE_LINKS, draw_feasible_configurations, path_offsets

    QC = np.deg2rad([0, -30, 0, 90, 0, 30, 0])
    for tag, T, B, guarded in (("4 dual_arm.py as shipped (T=50), per arm", 50, 8192, False),
                               ("4 synthetic: T=100 + joint limits + 4x6 sphere clearances, per arm", 100, 1024, True),
                               ("4 synthetic, 256 arms (one GPU's share of BASELINE's 1024 dual-arm problems over 8 GPUs)", 100, 256, True),
                               ("4 synthetic, large batch (guarded handles are compacted while they drain)", 100, 32768, True)):
        arm = RobotModel.builtin("kuka_lwr", time_derivs=[0, 1], name="kukal")
        arm.add_base_frame("global_world", xyz=[0.0, -0.25, 0.0])
        g = None
        if guarded:
            g = _lib.oh_guards()
            g.limits = 1
            for j in range(7):
                g.q_lo[j], g.q_up[j] = arm.lower_actuated_joint_limits[j], arm.upper_actuated_joint_limits[j]
            g.n_links, g.n_obstacles = 4, 6
            for l, (k, off) in enumerate(arm.link_attachments("end_effector_ball", SPHERE_LINKS)):
                g.link_joint[l] = k
                for i in range(3):
                    g.link_offset[l][i] = off[i]
        be = FigureEightBackend(arm.kinematic_chain("end_effector_ball"), T, 10.0 / (T - 1), path_offsets(T, [-0.1, 0.1, -0.2], [0.0, 0.0, 0.3]).T,
                                w_path=1.0, w_vel=0.01, max_iter=400, lock_orientation=False, fix_dq0=False, path_in_frame=False, guards=g)
        qc = draw_feasible_configurations(rng, B, arm, link_radius=0.1) if guarded else QC + rng.uniform(-0.1, 0.1, (B, 7))
        p = qc
        if guarded:
            obs_row = np.concatenate([[0.55, 0.0, 0.1 * (i + 1), 0.1] for i in range(6)])
            p = np.concatenate([qc, np.full((B, 4), 0.1), np.tile(obs_row, (B, 1))], 1)
        x0 = np.concatenate([np.tile(qc, (1, T)), np.zeros((B, 7 * (T - 1)))], 1)
        r = timed(be, np.ascontiguousarray(x0), np.ascontiguousarray(p))
        out.append({"config": tag, "batch": B, "T": T, "solves_per_s": B / r["ms"] * 1e3, "compactions": be.timing()["compactions"], **r})
        be.close()
    # ---- config 5: torque MPC, RNEA dynamics as equality rows (med7, T = 30), effort limit 58 N m so that the rows bind in part of the batch ----
    med7 = RobotModel.builtin("med7")
    link, T, dt = "lbr_link_ee", 30, 0.1
    qn = np.deg2rad([0, 30, 0, -90, 0, -30, 0])
    ts = np.arange(T) * dt
    loc = np.stack([0.2 * np.sin(ts * np.pi * 0.5), 0.1 * np.sin(ts * np.pi), np.zeros(T)])
    for B in (8192, 1024):
        qc = qn + rng.uniform(-0.1, 0.1, (B, 7))
        pose, _ = med7._kin(link).fk_jac(qc, want_jac=False)
        x, y, z, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
        Re = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                       np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                       np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
        goal = pose[:, None, :3] + np.einsum("bij,jt->bti", Re, loc)
        p = np.concatenate([qc, np.zeros((B, 7)), goal.reshape(B, -1)], 1)
        x0 = np.zeros((B, 4 * 7 * T))
        x0[:, : 7 * T] = np.tile(qc, (1, T))
        be = TorqueBackend(med7.kinematic_chain(link), med7.dynamics_tables(), T=T, dt=dt, w_path=1000.0, w_vel=0.1, w_tau=1e-4, tau_lo=-58.0, tau_up=58.0, max_iter=600)
        r = timed(be, np.ascontiguousarray(x0), np.ascontiguousarray(p))
        tm = be.timing()
        out.append({"config": "5 torque-control MPC, RNEA dynamics equality rows + effort limits (med7, T=30)", "batch": B, "T": T,
                    "solves_per_s": B / r["ms"] * 1e3, "iterations_launched": tm["iterations_launched"],
                    "instance_iterations": tm["work_instances"], "us_per_instance_iteration": r["ms"] * 1e3 / max(1.0, tm["work_instances"]), **r})
        be.close()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
