#!/bin/bash
# rocprofv3 kernel trace of repeated B = 1 solves (tools/gpu_latency.py): which launches make up the single-instance latency
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_lat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/tools/gpu_latency.py > $OUT/lat.log 2> $OUT/trace.log
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, sqlite3
c = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*.db", recursive=True)[0])
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(name[:60], calls, round(avg, 2), "us avg", round(pct, 1), "%")  # the top_kernels view is in microseconds
PY
tail -3 $OUT/lat.log
