#!/bin/bash
# rocprofv3 kernel trace of the torque family at B = 8192 (config 5): per-kernel time, gpurun_out/tq_prof/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/tq_prof -o tq -- python tools/gpu_tq_ipm_probe.py 8192 > gpurun_out/tq_prof.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tq_prof/**/*kernel_stats.csv", recursive=True)
print(f)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
