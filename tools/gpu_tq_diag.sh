#!/bin/bash
# Diagnosis of the torque kernels: per-launch durations by grid size and stall / memory counters (gpurun_out/tq_diag/).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/tq_diag; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/trace -o tq -- python $REPO/tools/gpu_tq_time.py 8192 > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o tq -- python $REPO/tools/gpu_tq_time.py 8192 > $OUT/pmc$i.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3, glob, collections, json, re
def short_name(n):
    m = re.search(r"(k_\\w+|__amd_\\w+)", n)
    return m.group(1) if m else n[:40]
out = {}
c = sqlite3.connect(glob.glob("$OUT/trace/*.db")[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
out["kernel_cols"] = cols
rows = c.execute("select name, grid_size_x, count(*), avg(end-start) from kernels group by name, grid_size_x order by name, grid_size_x desc").fetchall() if "grid_size_x" in cols else []
out["by_grid"] = [[short_name(n), g, k, a / 1e3] for n, g, k, a in rows if "k_tq" in n][:200]
pm = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob("$OUT/pmc*/")):
    try:
        p = sqlite3.connect(glob.glob(d + "*.db")[0])
        for n, cn, v in p.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
            pm[short_name(n)][cn] = v
        first = p.execute("select counter_name from counters_collection limit 1").fetchone()[0]
        for n, t in p.execute("select kernel_name, sum(duration) from counters_collection where counter_name = ? group by kernel_name", (first,)):
            pm[short_name(n)]["dur_ns_" + d.rstrip("/").split("/")[-1]] = t
    except Exception as e:
        out["err_" + d] = repr(e)
out["pmc"] = {k: dict(v) for k, v in pm.items() if k.startswith("k_tq")}
json.dump(out, open("$OUT/diag.json", "w"), indent=1)
print(json.dumps(out["pmc"], indent=1))
for r in out["by_grid"][:60]: print(r)
PY
