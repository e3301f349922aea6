#!/bin/bash
# kernel-level times of the torque-MPC family for both evaluation kernels (rocprofv3 --kernel-trace --stats), on one box
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  OH_TQ_EVAL3=$m rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/tq_prof_$m -o tq -- python $REPO/tools/gpu_torque_ab.py child > $REPO/gpurun_out/tq_prof_$m.json 2> $REPO/gpurun_out/tq_prof_$m.log
  echo "== OH_TQ_EVAL3=$m"; python - <<PY
import csv,glob
f=glob.glob("$REPO/gpurun_out/tq_prof_$m/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:6]:
    print(r["Name"][:50], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
done
