#!/bin/bash
# rocprofv3 kernel trace of the velocity-IK tick loop (tools/gpu_qp_rates.py): which kernels a B = 1 solve of the QP family spends its time in
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_qp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/tools/gpu_qp_rates.py > $OUT/qp.log 2> $OUT/trace.log
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, sqlite3
c = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*.db", recursive=True)[0])
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(name[:90], calls, round(avg, 1), "us avg", round(tot / 1e3, 1), "ms total", round(pct, 1), "%")
PY
tail -1 $OUT/qp.log
