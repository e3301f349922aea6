#!/bin/bash
# per-launch kernel durations of one velocity-limited batch -> gpurun_out/vel_trace.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vel_trace -- python $R/tools/gpu_vel_trace.py ${1:-16384} > $R/gpurun_out/vel_trace.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/vel_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
seq = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    seq.append((n, d, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
with open("$R/gpurun_out/vel_trace.txt", "w") as o:
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"{n:40s} {c:6d} launches {t/1e3:9.2f} ms  avg {t/c:8.1f} us\n")
    o.write("\nsequence of k_step_lg launches (us, grid):\n")
    o.write(" ".join(f"{d:.0f}/{g}" for n, d, g in seq if "k_step_lg" in n) + "\n")
    o.write("\nsequence of k_eval_lg launches (us):\n")
    o.write(" ".join(f"{d:.0f}" for n, d, g in seq if "k_eval_lg" in n) + "\n")
    o.write("\nsequence of k_couple_vel launches (us):\n")
    o.write(" ".join(f"{d:.0f}" for n, d, g in seq if "k_couple_vel" in n) + "\n")
PY
head -30 $R/gpurun_out/vel_trace.txt
