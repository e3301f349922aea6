#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/inv_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OH_DEBUG_OPTIONS="batch_invariant=1" rocprofv3 --kernel-trace --stats -d $OUT/trace -o inv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $OUT/bench.json 2> $OUT/log.txt
cd $REPO
python - <<PY
import sqlite3, glob, re, json
db = glob.glob("$OUT/trace/*.db")[0]
c = sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'").fetchall()]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kt} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
for n,k,t,a in rows[:16]:
    m=re.search(r"(k_\w+|oh_spec_\w+|__amd_\w+)", n); print(f"{(m.group(1) if m else n[:40]):28s} {k:6d} {t/1e6:9.2f} ms {a/1e3:9.1f} us {100*t/tot:5.1f}%")
j=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["compactions_per_step"], j["roofline"]["launches"])
PY
