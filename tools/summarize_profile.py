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
import re
import sqlite3
import subprocess
import sys

NAMES = ("k_tail", "k_tq_eval3", "k_tq_eval", "k_tq_step", "k_step_zc", "k_evalb_zc", "k_tq_setup", "k_tq_finalize", "k_tq_list", "k_retract", "k_evalb", "k_eval", "k_couple", "k_step", "k_fk_jac", "k_setup", "k_finalize", "k_compact_gather", "k_compact_scatter", "k_carry_gather", "k_carry_scatter", "k_scan_count", "k_scan_offsets", "k_scan_assign")


# the run-time specialised kernels (optas_amd/csrc/oh_jit.hip) appear under their own names; they are the same kernels compiled for one chain
SPEC = {"oh_spec_retract": "k_retract", "oh_spec_evalb_zc": "k_evalb_zc", "oh_spec_evalb": "k_evalb", "oh_spec_tail": "k_tail", "oh_spec_fk_soa": "k_fk_jac", "oh_spec_fk_aos": "k_fk_jac"}


def short(name: str) -> str:
    for k, v in SPEC.items():
        if k in name:
            return v
    for k in NAMES:
        if k in name:
            return k
    return name[:48]


def co_key(mangled: str):
    """'k_step' for the ndof-7 instantiation of k_step (k_fk_jac: the SoA, 7-joint one), 'k_step<6>' for ndof 6, None for the other variants."""
    m = re.match(r"_Z(?:N12_GLOBAL__N_1)?(\d+)", mangled)
    if not m:
        return None
    n = int(m.group(1))
    base = mangled[m.end() : m.end() + n]
    rest = mangled[m.end() + n :]
    if base == "k_fk_jac":
        return base if rest.startswith("ILb1ELi7E") else None
    if rest.startswith("ILi7E") or not rest.startswith("I"):
        return base
    if rest.startswith("ILi6E"):
        return base + "<6>"
    return None


def code_object_registers():
    """Registers of every kernel as the code object states them (.vgpr_count / .agpr_count / scratch / LDS of the AMDGPU metadata): the
    kernel table of rocprofv3's database under-reports unified-register kernels by 2x (round-1 verdict).  The shared library is
    unbundled with clang-offload-bundler and its notes read with llvm-readelf."""
    so = os.path.join("optas_amd", "liboptas_hip.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    try:
        import struct

        blob = open(so, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        tmp = "/tmp/oh_co_%d" % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        pos, n = blob.find(magic), 0
        while pos >= 0:  # one bundle per translation unit in .hip_fatbin
            (cnt,) = struct.unpack_from("<Q", blob, pos + 24)
            off = pos + 32
            for _ in range(cnt):
                eoff, esz, tsz = struct.unpack_from("<QQQ", blob, off)
                triple = blob[off + 24 : off + 24 + tsz].decode()
                off += 24 + tsz
                if "gfx950" in triple and esz:
                    path = f"{tmp}/dev{n}.co"
                    open(path, "wb").write(blob[pos + eoff : pos + eoff + esz])
                    n += 1
                    txt = subprocess.run([f"{llvm}/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
                    for blk in txt.split("- .agpr_count:")[1:]:
                        g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, None])[1]
                        name = g("name")
                        key = co_key(name) if name else None
                        if key:
                            out[key] = {
                                "agpr": int(blk.split()[0]), "vgpr": int(g("vgpr_count") or 0), "sgpr": int(g("sgpr_count") or 0),
                                "scratch": int(g("private_segment_fixed_size") or 0), "lds": int(g("group_segment_fixed_size") or 0),
                                "spilled_vgprs": int(g("vgpr_spill_count") or 0)}
            pos = blob.find(magic, pos + 24)
        # code objects of the specialised kernels (in-tree cache filled by __graft_entry__.build() / the run itself): they replace the
        # generic entries, since they are what the profiled run launched (bench.py says so in "specialized_kernels")
        for path in sorted(glob.glob(os.path.join(".optas_hip_cache", "spec_*.hsaco")), key=os.path.getmtime):
            txt = subprocess.run([f"{llvm}/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, None])[1]
                name = g("name")
                key = SPEC.get(name) if name != "oh_spec_fk_aos" else None
                if key:
                    out[key] = {
                        "agpr": int(blk.split()[0]), "vgpr": int(g("vgpr_count") or 0), "sgpr": int(g("sgpr_count") or 0),
                        "scratch": int(g("private_segment_fixed_size") or 0), "lds": int(g("group_segment_fixed_size") or 0),
                        "spilled_vgprs": int(g("vgpr_spill_count") or 0), "specialised": True}
    except Exception as e:  # pragma: no cover
        print("code object registers unavailable:", e)
    return out


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
            pair = {("k_evalb" if n == "k_evalb_zc" else n): (calls, tot) for n, calls, tot in ((short(n), c_, t_) for n, c_, t_, _, _ in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels")) if n in ("k_retract", "k_evalb", "k_evalb_zc")}
            if len(pair) == 2:
                calls = pair["k_retract"][0]
                tot = pair["k_retract"][1] + pair["k_evalb"][1]
                w.writerow(["# k_eval = the pair k_retract + k_evalb[_zc] (one evaluation of the trial knots; what bench.py times as k_eval)"])
                w.writerow(["k_eval (pair)", calls, f"{tot:.1f}", f"{tot / max(calls, 1):.2f}", ""])
            w.writerow([])
            w.writerow(["# per grid size (x dimension = instances still in the batch, or units for k_fk_jac)"])
            w.writerow(["kernel", "grid_x", "calls", "avg_us", "min_us", "max_us", "vgpr", "agpr", "sgpr", "scratch"])
            co = code_object_registers()
            agg = collections.defaultdict(list)
            info = {}
            for n, gx, d, v, a, sg, sc in c.execute("select name,grid_x,duration,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size from kernels"):
                k = short(n)
                if k in NAMES:
                    agg[(k, gx)].append(d)
                    info[k] = (co[k]["vgpr"], co[k]["agpr"], co[k]["sgpr"], co[k]["scratch"]) if k in co else (v, a, sg, sc)
            for (k, gx) in sorted(agg):
                d = agg[(k, gx)]
                w.writerow([k, gx, len(d), f"{sum(d)/len(d)/1e3:.1f}", f"{min(d)/1e3:.1f}", f"{max(d)/1e3:.1f}", *info[k]])
            w.writerow([])
            w.writerow(["# registers above are read from the code object of optas_amd/liboptas_hip.so (llvm-readelf --notes), not from rocprofv3's kernel table"])
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
        if "k_step_zc" in summary and "k_step" not in summary:  # bench.py looks the sweep up as k_step whichever variant ran
            summary["k_step"] = dict(summary["k_step_zc"], note="k_step_zc: the sweep with the coupling folded in")
        if "k_evalb_zc" in summary and "k_evalb" not in summary:
            summary["k_evalb"] = dict(summary["k_evalb_zc"], note="k_evalb_zc")
        if "k_eval" not in summary and "k_retract" in summary and "k_evalb" in summary:
            a, b = summary["k_retract"], summary["k_evalb"]
            summary["k_eval"] = {k: (a[k] + b[k]) if isinstance(a.get(k), (int, float)) and isinstance(b.get(k), (int, float)) else None for k in a}
            summary["k_eval"]["note"] = "k_retract + k_evalb: the two launches of one trial-knot evaluation"
        json.dump(
            {"tag": tag, "note": "average per launch over the profiled bench run (all batch sizes the run went through); FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": pmc, "summary": summary},
            open(f"profiles/{tag}_pmc.json", "w"),
            indent=1,
        )
        # what bench.py needs to rescale the traffic to its own run: the units per launch of the profiled run and where the numbers came from
        meta = {"tag": tag}
        try:
            meta["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except Exception:
            meta["commit"] = None
        for n in ("bench_fetch.json", "bench_trace.json"):
            pth = os.path.join(root, n)
            if os.path.exists(pth) and os.path.getsize(pth) > 0:
                try:
                    bj = json.loads(open(pth).read().strip().splitlines()[-1])
                    meta["units_per_launch"] = bj["roofline"]["units_per_launch_avg"]
                    meta["fk_units"] = bj["roofline_fk_jac"]["units"]
                    break
                except Exception:
                    pass
        for k in summary:
            summary[k]["units_per_launch"] = meta.get("fk_units") if k == "k_fk_jac" else meta.get("units_per_launch")
        summary["_meta"] = meta
        json.dump(summary, open("profiles/pmc_traffic.json", "w"), indent=1)
        print("pmc ->", f"profiles/{tag}_pmc.json")
    for n in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(root, n)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            open(f"profiles/{tag}_{n}", "w").write(open(p).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
