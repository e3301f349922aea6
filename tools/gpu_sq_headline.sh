#!/bin/bash
# SQ-counter passes over the headline workload (bench.py --timed-only): f64 instruction mix, busy / wait cycles and LDS / memory instruction counts of
# k_retract, k_evalb_zc, k_step_zc, k_tail, k_carry_gather (round-4 verdict, Next 4: "0.55 is the ceiling" was asserted, not shown).
# Own passes, counters only with --kernel-trace (never with sys / hip / hsa tracing).  Output: gpurun_out/${TAG:-r06}_sq.json (copy into profiles/).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_headline; rm -rf $OUT; mkdir -p $OUT
ARGS=${1:-"--steps 1 --warmup 1 --no-cpu-baseline --timed-only"}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o sq -- python $REPO/bench.py $ARGS > $OUT/bench$i.json 2> $OUT/pmc$i.log
done
cd $REPO
python - <<PY
import sqlite3, glob, collections, json, re
OUT = "$OUT"
def short_name(n):
    m = re.search(r"(oh_spec_\\w+|k_\\w+|__amd_\\w+)", n)
    return m.group(1) if m else n[:48]
pm = collections.defaultdict(lambda: collections.defaultdict(float))
errs = {}
for d in sorted(glob.glob(OUT + "/pmc*/")):
    try:
        p = sqlite3.connect(glob.glob(d + "*.db")[0])
        for n, cn, v, k in p.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            pm[short_name(n)][cn] = v
            pm[short_name(n)]["launches"] = k
        first = p.execute("select counter_name from counters_collection limit 1").fetchone()[0]
        for n, t in p.execute("select kernel_name, sum(duration) from counters_collection where counter_name = ? group by kernel_name", (first,)):
            pm[short_name(n)]["dur_ns_" + d.rstrip("/").split("/")[-1]] = t
    except Exception as e:
        errs[d] = repr(e)
res = {}
for k, v in pm.items():
    if not (k.startswith("k_") or k.startswith("oh_spec")):
        continue
    v = dict(v)
    fma, mul, add, tr = v.get("SQ_INSTS_VALU_FMA_F64", 0), v.get("SQ_INSTS_VALU_MUL_F64", 0), v.get("SQ_INSTS_VALU_ADD_F64", 0), v.get("SQ_INSTS_VALU_TRANS_F64", 0)
    secs = v.get("dur_ns_pmc1", 0) * 1e-9
    d = {"counters": v}
    if secs > 0:
        flop = 64.0 * (2 * fma + mul + add)
        d["f64_tflops_all_lanes"] = flop / secs * 1e-12
        d["f64_frac_of_78.6_TF"] = flop / secs * 1e-12 / 78.6
    if v.get("SQ_INSTS_VALU"):
        d["f64_share_of_valu"] = (fma + mul + add + tr) / v["SQ_INSTS_VALU"]
    if v.get("SQ_WAVE_CYCLES"):
        d["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"]
        d["any_inst_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
    res[k] = d
json.dump({"what": "rocprofv3 --pmc passes over bench.py $ARGS; counters summed over every launch of a kernel and all SEs/XCDs as rocprofv3 reports them; f64 flop = 64 lanes x (2 FMA + MUL + ADD) wave instructions",
           "kernels": res, "errors": errs}, open("$REPO/gpurun_out/${TAG:-r06}_sq.json", "w"), indent=1)
for k in ("oh_spec_retract", "oh_spec_evalb_zc", "k_step_zc", "oh_spec_tail", "k_carry_gather"):
    if k in res:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in res[k].items() if a != "counters"})
print("errors", errs)
PY
# the raw databases of six passes are ~100 MB: gpurun merges at most 64 MiB back -- keep the condensed file and one pass's logs only
rm -rf $OUT/pmc*/ 2>/dev/null
ls -la $REPO/gpurun_out/${TAG:-r06}_sq.json
