#!/bin/bash
# per-launch durations of the torque kernels for one instance (latency floor), with and without the Anderson acceleration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for aa in 3 0; do
cat > /tmp/tq_b1.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from tools.bench_configs import run_configs
print(run_configs(sample=0, only="torque", torque_batches=(2,)))
PY
OH_TQ_AA=$aa rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tq_b1_$aa -- python /tmp/tq_b1.py > $R/gpurun_out/tq_b1_$aa.log 2>&1
python - <<PY
import csv, glob, statistics as st
f = sorted(glob.glob("$R/gpurun_out/tq_b1_$aa/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tq_eval" in r["Kernel_Name"]]
sp = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tq_step" in r["Kernel_Name"]]
print("OH_TQ_AA=$aa: launches", len(sp), "eval median", st.median(ev), "step median", st.median(sp), "step min/max", min(sp), max(sp), "last 40 step median", st.median(sp[-40:]))
PY
done
