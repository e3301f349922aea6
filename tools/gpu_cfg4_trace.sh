#!/bin/bash
# per-kernel durations of config 4 synthetic -> gpurun_out/cfg4_trace_<B>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-256}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cfg4_trace -- python $R/tools/gpu_cfg4_trace.py $B > $R/gpurun_out/cfg4_trace_$B.log 2>&1
python - <<PY
import csv, glob, collections
f = sorted(glob.glob("$R/gpurun_out/cfg4_trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve only: from the last k_setup on
idx = max(i for i, r in enumerate(rows) if "k_setup" in r["Kernel_Name"] and "guards" not in r["Kernel_Name"])
rows = rows[idx:]
agg = collections.OrderedDict()
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = 0
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
with open("$R/gpurun_out/cfg4_trace_$B.txt", "w") as o:
    o.write(f"last solve: span {(t1-t0)/1e6:.2f} ms, kernels busy {busy/1e3:.2f} ms, gaps {(t1-t0)/1e6-busy/1e3:.2f} ms, {len(rows)} launches\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{n:40s} {c:6d} launches {t/1e3:9.2f} ms  avg {t/c:8.1f} us\n")
PY
cat $R/gpurun_out/cfg4_trace_$B.log | tail -3
cat $R/gpurun_out/cfg4_trace_$B.txt
