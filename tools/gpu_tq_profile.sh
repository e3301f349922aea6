#!/bin/bash
# Torque family (config 5) at B = 8192: rocprofv3 kernel trace (pass 1) and f64 VALU instruction counters (pass 2, counters only + kernel trace).
# -> gpurun_out/profiles/${TAG:-r05}_torque_kernel_stats.csv, ${TAG:-r05}_torque_pmc.json (copy into profiles/ and commit)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/tq_prof; rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o tq -- python $REPO/tools/gpu_tq_ipm_probe.py 8192 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU -d $OUT/pmc -o tq -- python $REPO/tools/gpu_tq_ipm_probe.py 8192 > $OUT/pmc.log 2>&1
cd $REPO
python - <<PY
import sqlite3, glob, collections, json, re
def short_name(n):
    m = re.search(r"(k_\\w+|__amd_\\w+)", n)
    return m.group(1) if m else n[:40]
db = glob.glob("$OUT/trace/*.db")[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open("$REPO/gpurun_out/profiles/${TAG:-r05}_torque_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats, tools/gpu_tq_ipm_probe.py 8192 (two solves of 8192 instances; config 5)\nkernel,calls,total_us,avg_us,pct\n")
    for n, k, t, a in rows:
        short = short_name(n)
        f.write(f"{short},{k},{t/1e3:.1f},{a/1e3:.2f},{100*t/tot:.2f}\n")
print(open("$REPO/gpurun_out/profiles/${TAG:-r05}_torque_kernel_stats.csv").read())
out = {}
try:
    p = sqlite3.connect(glob.glob("$OUT/pmc/*.db")[0])
    rows = p.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    dur = dict(p.execute("select kernel_name, sum(duration) from counters_collection where counter_name = 'SQ_INSTS_VALU' group by kernel_name").fetchall())
    agg = collections.defaultdict(dict)
    for n, cn, v, k in rows:
        agg[n][cn] = v
    for n, d in agg.items():
        short = short_name(n)
        if not short.startswith("k_tq"): continue
        fma, add, mul = d.get("SQ_INSTS_VALU_FMA_F64", 0), d.get("SQ_INSTS_VALU_ADD_F64", 0), d.get("SQ_INSTS_VALU_MUL_F64", 0)
        flop = 64.0 * (2 * fma + add + mul)
        t = dur.get(n, 0) / 1e9
        out[short] = {"wave_insts": d, "f64_flop_if_all_lanes_active": flop, "seconds": t, "tflops": flop / t / 1e12 if t else None,
                      "frac_of_78.6_TF": flop / t / 78.6e12 if t else None, "f64_share_of_valu": (fma + add + mul) / max(d.get("SQ_INSTS_VALU", 1), 1)}
except Exception as e:
    out["error"] = repr(e)
json.dump(out, open("$REPO/gpurun_out/profiles/${TAG:-r05}_torque_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
