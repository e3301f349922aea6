#!/bin/bash
# one rocprofv3 PMC pass with an arbitrary counter list: tools/pmc_probe.sh "<counters>" <tag> [bench args]
set -u
CNT="$1"; TAG=$2; ARGS=${3:-"--steps 1 --warmup 1 --no-cpu-baseline"}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT -d $OUT -o pmc -- python $REPO/bench.py $ARGS > $OUT/bench.json 2> $OUT/log.txt
cd $REPO
python - <<PY
import sqlite3, glob, collections
c=sqlite3.connect(glob.glob("$OUT/*.db")[0])
rows=c.execute("select kernel_name, grid_size_x, counter_name, value, duration from counters_collection").fetchall()
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int); dur=collections.defaultdict(float)
for n,gx,cn,v,d in rows:
    k=[s for s in ("k_eval","k_couple","k_step","k_tail","k_fk_jac","k_move") if s in n]
    if not k: continue
    key=(k[0],gx)
    agg[key][cn]+=v
    if cn==rows[0][2] or True: pass
for n,gx,cn,v,d in rows:
    k=[s for s in ("k_eval","k_couple","k_step","k_tail","k_fk_jac","k_move") if s in n]
    if not k: continue
    key=(k[0],gx)
    if cn=="$CNT".split()[0]: cnt[key]+=1; dur[key]+=d
for key in sorted(agg):
    if cnt[key]==0: continue
    print(key, "launches",cnt[key], "avg_us %.1f"%(dur[key]/cnt[key]/1e3), {k: "%.3g"%(v/cnt[key]) for k,v in agg[key].items()})
PY
