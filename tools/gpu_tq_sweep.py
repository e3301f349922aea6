"""Config 5 on the driver's instance set (tools/bench_configs.py) under settings of the curvature switch: python tools/gpu_tq_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_configs import run_configs
    o = run_configs(sample=0, torque_batches=(8192, 1024)) if os.environ.get("FULL") else run_configs(sample=0, torque_batches=(8192,), only="torque")
    for key in ("config5_torque_b8192", "config5_torque_b1024"):
        v = o.get(key)
        if v: print(key, json.dumps({k: v.get(k) for k in ("device_ms", "iterations_launched", "iters_p50", "iters_p90", "iters_max", "converged_frac")}))
else:
    for cf, ca in ((("0.1", "3"), ("0.03", "4"), ("0.01", "4"), ("1e-9", "2"), ("0.03", "3")) if os.environ.get("FULL") else (("0.1", "3"), ("0.03", "4"), ("0.01", "4"), ("1e-9", "2"), ("0.03", "3"), ("0.1", "4"), ("0.3", "3"), ("1e-9", "3"))):
        env = dict(os.environ, OH_DEBUG_OPTIONS=f'tq_curv_from={cf},tq_curv_after={ca}')
        out = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
        print(cf, ca, " | ".join((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-2:]), flush=True)
