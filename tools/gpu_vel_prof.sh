#!/bin/bash
# config 2 + joint-velocity limits, B = 65536: kernel totals of one warm solve pair (gpurun_out/vel_prof/)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/vel_prof; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/vel_one.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
from examples.figure_eight_plan import setup_solver
B = int(os.environ.get("B", 65536))
qcs = np.deg2rad([0, 30, 0, -90, 0, -30, 0])[None] + np.random.default_rng(20240607 + 2).uniform(-0.1, 0.1, (B, 7))
kuka, solver = setup_solver(velocity_limits=True, solver_options={"max_iter": 600, "tol": 1e-6})
x0 = np.zeros((B, solver.opt.nx)); x0[:, :350] = np.repeat(qcs, 50, axis=0).reshape(B, 350)
be = solver.backend
for rep in range(3):
    r = be.solve(x0, np.ascontiguousarray(qcs))
    it = np.asarray(r.iters); t = be.timing()
    print("B", B, "ms", round(t["solve_ms"], 2), "launched", t["iterations_launched"], "conv", float((np.asarray(r.status) == 0).mean()), "p50", np.median(it), "p90", np.percentile(it, 90), "p99", np.percentile(it, 99), "p99.9", np.percentile(it, 99.9), "max", it.max(), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o v -- python /tmp/vel_one.py > $OUT/log.txt 2>&1
cd $REPO
grep "^B" $OUT/log.txt
python - <<PY
import sqlite3, glob, re
c = sqlite3.connect(glob.glob("$OUT/*.db")[0])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), max(end-start), max(vgpr_count), max(scratch_size), max(lds_size), max(grid_x) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for n, k, t, a, mx, v, sc, lds, gx in rows[:16]:
    m = re.search(r"(k_\\w+|__amd_\\w+)", n)
    print("  %-22s calls %5d total %8.2f ms %5.1f%% avg %7.1f max %8.1f us  vgpr %3d scratch %5d lds %6d grid %8d" % (m.group(1) if m else n[:30], k, t / 1e6, 100.0 * t / tot, a/1e3, mx / 1e3, v, sc, lds, gx))
PY
