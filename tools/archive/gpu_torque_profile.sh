#!/bin/bash
# rocprofv3 kernel trace of the torque-MPC batch (tools/gpu_torque_b8192.py, TQ_B instances): per-kernel launch durations
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_tq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/tools/gpu_torque_b8192.py > $OUT/tq.log 2> $OUT/trace.log
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, sqlite3
c = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*.db", recursive=True)[0])
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(name[:70], calls, round(avg, 1), "us avg", round(tot / 1e3, 1), "ms total", round(pct, 1), "%")  # the top_kernels view is in microseconds
PY
tail -2 $OUT/tq.log
