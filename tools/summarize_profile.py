"""Condense rocprofv3 output (rocpd SQLite: kernel-trace stats + PMC passes) into small committed files.

  python tools/summarize_profile.py gpurun_out/prof_r01 r01
writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, plus a per-grid-size
break-down), profiles/<tag>_pmc.json and refreshes profiles/pmc_traffic.json (the per-launch HBM bytes
that bench.py reports as roofline.traffic).

FETCH_SIZE / WRITE_SIZE are rocprofv3 derived counters in KiB.  Per MI355X_MICROARCH.md (HBM section) the
gfx950 FETCH_SIZE counts 128-B read requests at 64 B, i.e. reports half of the bytes of a wide coalesced
stream: the read side is doubled ("corrected"); raw values are kept next to it.  WRITE_SIZE is used as is.
"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

NAMES = ("k_retract", "k_evalb", "k_eval", "k_couple", "k_step", "k_fk_jac", "k_setup", "k_finalize", "k_compact_gather", "k_compact_scatter", "k_carry_gather", "k_carry_scatter", "k_scan_count", "k_scan_offsets", "k_scan_assign")


def short(name: str) -> str:
    for k in NAMES:
        if k in name:
            return k
    return name[:48]


def db(root, sub):
    out = glob.glob(os.path.join(root, sub, "*.db"))
    return sqlite3.connect(out[0]) if out else None


def main(root, tag):
    os.makedirs("profiles", exist_ok=True)
    c = db(root, "trace")
    if c is not None:
        with open(f"profiles/{tag}_kernel_stats.csv", "w") as fh:
            w = csv.writer(fh)
            w.writerow(["# rocprofv3 --kernel-trace --stats (top_kernels view)"])
            w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
            for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
                w.writerow([short(name), calls, f"{tot:.1f}", f"{avg:.2f}", f"{pct:.2f}"])
            pair = {n: (calls, tot) for n, calls, tot in ((short(n), c_, t_) for n, c_, t_, _, _ in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels")) if n in ("k_retract", "k_evalb")}
            if len(pair) == 2:
                calls = pair["k_retract"][0]
                tot = pair["k_retract"][1] + pair["k_evalb"][1]
                w.writerow(["# k_eval = the pair k_retract + k_evalb (one evaluation of the trial knots; what bench.py times as k_eval)"])
                w.writerow(["k_eval (pair)", calls, f"{tot:.1f}", f"{tot / max(calls, 1):.2f}", ""])
            w.writerow([])
            w.writerow(["# per grid size (x dimension = instances still in the batch, or units for k_fk_jac)"])
            w.writerow(["kernel", "grid_x", "calls", "avg_us", "min_us", "max_us", "vgpr", "agpr", "sgpr", "scratch"])
            agg = collections.defaultdict(list)
            info = {}
            for n, gx, d, v, a, sg, sc in c.execute("select name,grid_x,duration,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size from kernels"):
                k = short(n)
                if k in NAMES:
                    agg[(k, gx)].append(d)
                    info[k] = (v, a, sg, sc)
            for (k, gx) in sorted(agg):
                d = agg[(k, gx)]
                w.writerow([k, gx, len(d), f"{sum(d)/len(d)/1e3:.1f}", f"{min(d)/1e3:.1f}", f"{max(d)/1e3:.1f}", *info[k]])
        print("kernel stats ->", f"profiles/{tag}_kernel_stats.csv")
    pmc = {}
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        c = db(root, sub)
        if c is None:
            continue
        acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
        for name, val, dur in c.execute("select name,counter_value,duration from pmc_events where counter_name=?", (cname,)):
            k = short(name)
            if k in NAMES:
                acc[k][0] += float(val)
                acc[k][1] += 1
                acc[k][2] += dur
        for k, (tot, n, dur) in acc.items():
            pmc.setdefault(k, {})[cname] = {"sum_kib": tot, "launches": n, "avg_kib_per_launch": tot / max(n, 1), "avg_us": dur / max(n, 1) / 1e3}
    if pmc:
        summary = {}
        for k, d in pmc.items():
            fs = d.get("FETCH_SIZE", {}).get("avg_kib_per_launch")
            ws = d.get("WRITE_SIZE", {}).get("avg_kib_per_launch")
            e = {"fetch_kib_raw": fs, "write_kib_raw": ws}
            if fs is not None and ws is not None:
                e["bytes_per_launch_raw"] = (fs + ws) * 1024.0
                e["bytes_per_launch"] = (2.0 * fs + ws) * 1024.0
            summary[k] = e
        if "k_eval" not in summary and "k_retract" in summary and "k_evalb" in summary:
            a, b = summary["k_retract"], summary["k_evalb"]
            summary["k_eval"] = {k: (a[k] + b[k]) if a.get(k) is not None and b.get(k) is not None else None for k in a}
            summary["k_eval"]["note"] = "k_retract + k_evalb: the two launches of one trial-knot evaluation"
        json.dump(
            {"tag": tag, "note": "average per launch over the profiled bench run (all batch sizes the run went through); FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": pmc, "summary": summary},
            open(f"profiles/{tag}_pmc.json", "w"),
            indent=1,
        )
        json.dump(summary, open("profiles/pmc_traffic.json", "w"), indent=1)
        print("pmc ->", f"profiles/{tag}_pmc.json")
    for n in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(root, n)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            open(f"profiles/{tag}_{n}", "w").write(open(p).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
