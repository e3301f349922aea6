#!/bin/bash
# config 4 synthetic: kernel totals + resource table + duration by grid (gpurun_out/cfg4_prof/)
set -u
REPO=$(pwd); B=${1:-1024}; OUT=$REPO/gpurun_out/cfg4_prof_$B; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o c4 -- python $REPO/tools/gpu_cfg4_trace.py $B > $OUT/log.txt 2>&1
cd $REPO
tail -3 $OUT/log.txt | grep "^B"
python - <<PY
import sqlite3, glob, re
c = sqlite3.connect(glob.glob("$OUT/*.db")[0])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
for n, k, t, a, mx, v, ag, sc, lds, gx, wx in rows[:14]:
    m = re.search(r"(k_\\w+|__amd_\\w+)", n)
    print("  %-22s calls %5d total %8.2f ms avg %7.1f max %7.1f us  vgpr %3d agpr %3d scratch %5d lds %6d grid %8d wg %4d" % (m.group(1) if m else n[:30], k, t / 1e6, a/1e3, mx / 1e3, v, ag, sc, lds, gx, wx))
PY
