#!/bin/bash
# rocprofv3 SQ counter pass (issue / stall breakdown per kernel); run on the GPU box through gpurun.  Counters only + kernel-trace.
set -u
TAG=${1:-sq}
ARGS=${2:-"--steps 1 --warmup 1 --no-cpu-baseline"}
CTRS=${3:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/sq -o sq -- python $REPO/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.log
cd $REPO
python - "$OUT" <<'PY'
import sys, glob, sqlite3, collections, json, re
root = sys.argv[1]
dbs = glob.glob(root + "/sq/**/*.db", recursive=True)
c = sqlite3.connect(dbs[0])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for name, cn, val, dur in c.execute("select name,counter_name,counter_value,duration from pmc_events"):
    k = re.sub(r"\(.*", "", name).replace("void ", "").strip()
    acc[k][cn] += float(val)
    acc[k]["_dur_ns_" + cn] += dur
    n[(k, cn)] += 1
out = {}
for k, d in acc.items():
    e = {cn: v for cn, v in d.items() if not cn.startswith("_")}
    first = next(iter(e))
    e["launches"] = n[(k, first)]
    e["avg_us"] = d["_dur_ns_" + first] / max(n[(k, first)], 1) / 1e3
    out[k] = e
json.dump(out, open(root + "/sq_summary.json", "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()})
PY
