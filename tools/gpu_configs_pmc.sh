#!/bin/bash
# f64 flop per work unit of the kernels of configs 1, 3, 4, 5 (the numerators of their roofline objects): one rocprofv3 counter pass per family
# (counters only + kernel trace), SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 per kernel, divided by the work units the same process reports.
# -> gpurun_out/profiles/${TAG:-r06}_configs_flops.json (copy to profiles/ AND to profiles/configs_flops.json, which tools/bench_configs.py reads)
set -u
REPO=$(pwd); TAG=${TAG:-r06}; OUT=$REPO/gpurun_out/cfgpmc; rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
for fam in ${FAMILIES:-ik pm pm_thread guarded torque}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 -d $OUT/$fam -o $fam -- python $REPO/tools/bench_configs.py --probe $fam > $OUT/$fam.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3, glob, collections, json, re, os
GROUPS = {"ik": [("k_ik", "k_ik (the whole solve: one launch)")], "pm": [("k_pm", "k_pm (the whole solve: one launch)")], "pm_thread": [("k_pm", "k_pm (the whole solve: one launch)")],
          "guarded": [("k_eval_guarded", "k_eval_guarded"), ("k_eval_free", "k_eval_guarded"), ("k_step_free", "k_step_free*"), ("k_step_guarded", "k_step_free*")],
          "torque": [("k_tq_eval3", "k_tq_eval3+k_tq_curv"), ("k_tq_curv", "k_tq_eval3+k_tq_curv"), ("k_tq_step", "k_tq_step")]}
out = {}
for fam, groups in GROUPS.items():
    try:
        units = json.load(open(f"$OUT/{fam}_units.json"))
        db = sqlite3.connect(glob.glob(f"$OUT/{fam}/*.db")[0])
        rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = {}
        for n, cn, v, k in rows:
            agg[n][cn] += v
            calls[n] = k
        flop, names = collections.defaultdict(float), collections.defaultdict(list)
        other = {}
        for n, d in agg.items():
            f = 64.0 * (2 * d.get("SQ_INSTS_VALU_FMA_F64", 0) + d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0))
            m = re.search(r"(k_\w+)", n)
            short = m.group(1) if m else n[:40]
            for pat, g in groups:
                if pat in short:
                    flop[g] += f
                    names[g].append(short)
                    break
            else:
                if f > 0: other[short] = f
        out[fam] = {"tag": "$TAG", "units": units["units"], "solves": units["solves"], "flop_per_unit": {g: v / units["units"] for g, v in flop.items()},
                    "kernels": {g: sorted(set(v)) for g, v in names.items()}, "f64_flop_total": dict(flop), "f64_flop_of_other_kernels": other}
    except Exception as e:
        out[fam] = {"error": repr(e)}
json.dump(out, open("$REPO/gpurun_out/profiles/${TAG}_configs_flops.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY
