#!/bin/bash
# A/B of two builds on one box with the per-grid-size table of k_step_zc (rocprofv3 kernel trace, one stream): tools/gpu_ab_grid.sh optas_amd/liboptas_hip_old.so
OLD=$(pwd)/$1; REPO=$(pwd); mkdir -p gpurun_out/abgrid
bash tools/gpu_ab_lib.sh $1 > gpurun_out/abgrid/ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
for tag in old new; do
  L=""; [ $tag = old ] && L=$OLD
  rm -rf $REPO/gpurun_out/abgrid/$tag
  OH_DEBUG_OPTIONS=streams=1 OPTAS_HIP_LIBRARY=$L rocprofv3 --kernel-trace -d $REPO/gpurun_out/abgrid/$tag -o t -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only > /dev/null 2>&1
done
cd $REPO; cat gpurun_out/abgrid/ab.log
python - <<'PY'
import sqlite3, glob, collections
def table(tag):
    db = glob.glob(f"gpurun_out/abgrid/{tag}/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'").fetchall()]
    kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
    out={}
    for pat in ("k_step_zc","retract","evalb_zc"):
        rows=c.execute(f"select d.grid_size_x, count(*), avg(d.end-d.start) from {kt} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' group by d.grid_size_x").fetchall()
        out[pat]={g:(n,a/1e3) for g,n,a in rows}
        tot=c.execute(f"select sum(d.end-d.start) from {kt} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%'").fetchone()[0]
        out[pat+"_total_ms"]=tot/1e6
    return out
o,n=table("old"),table("new")
for pat in ("k_step_zc","retract","evalb_zc"):
    print(pat, "total ms old", round(o[pat+"_total_ms"],2), "new", round(n[pat+"_total_ms"],2))
gs=sorted(set(o["k_step_zc"])&set(n["k_step_zc"]), reverse=True)[:26]
for g in gs: print("k_step grid", g, "old", round(o["k_step_zc"][g][1]), "new", round(n["k_step_zc"][g][1]))
PY
python - <<'PY'
import sqlite3, glob, re
for tag in ("old","new"):
    db = glob.glob(f"gpurun_out/abgrid/{tag}/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'").fetchall()]
    kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
    rows=c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kt} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    print(tag, "all kernels ms", round(sum(r[2] for r in rows)/1e6,2))
    for n,k,t in rows[3:11]:
        m=re.search(r"(k_\w+|oh_spec_\w+|__amd_\w+)", n); print(f"   {(m.group(1) if m else n[:30])[:28]:28s} {k:5d} {t/1e6:8.2f} ms {t/k/1e3:8.1f} us")
PY
