#!/bin/bash
# kernel totals of the torque family at B = 8192 (4 solves) for the environment given: tools/gpu_tq_trace.sh [tag]
set -u
REPO=$(pwd); TAG=${1:-default}; OUT=$REPO/gpurun_out/tq_trace_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o tq -- python $REPO/tools/gpu_tq_time.py 8192 > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import sqlite3, glob, re
c = sqlite3.connect(glob.glob("$OUT/*.db")[0])
rows = c.execute("select name, count(*), sum(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
print("$TAG")
for n, k, t, mx in rows[:6]:
    m = re.search(r"(k_\\w+|__amd_\\w+)", n)
    print("  %-16s calls %5d total %9.2f ms  max %8.1f us" % (m.group(1) if m else n[:30], k, t / 1e6, mx / 1e3))
PY
