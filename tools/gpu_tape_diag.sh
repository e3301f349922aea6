#!/bin/bash
# Counters of the wavefront-per-instance tape kernel on the planner (4 instances): instructions and waits per wavefront (gpurun_out/tape_diag/)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/tape_diag; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o t -- python $REPO/tools/gpu_tape_wave.py wave > $OUT/pmc$i.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3, glob, collections, json
pm = collections.defaultdict(float)
for d in sorted(glob.glob("$OUT/pmc*/")):
    try:
        p = sqlite3.connect(glob.glob(d + "*.db")[0])
        for n, cn, v in p.execute("select kernel_name, counter_name, sum(value) from counters_collection where kernel_name like '%k_tape_wave%' group by kernel_name, counter_name"):
            pm[cn] = v
        for n, cnt, t in p.execute("select name, count(*), sum(end-start) from kernels where name like '%k_tape_wave%' group by name"):
            pm["launches"] = cnt; pm["dur_ns_" + d.rstrip("/").split("/")[-1]] = t
    except Exception as e:
        pm["err_" + d] = repr(e)
print(json.dumps(pm, indent=1))
PY
grep "wave B" $OUT/pmc1.log | tail -1 | cut -c1-80
