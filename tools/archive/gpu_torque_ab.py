"""A/B of the torque-MPC evaluation kernels on one box: OH_TQ_EVAL3=1 (one lane per joint, three tangents per primal) against 0 (one lane per
tangent direction, round 2), BASELINE configs[4] at B = 8192 / 1024 / 1.  python tools/gpu_torque_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs

    out = bench_configs.run_configs(sample=0, only="torque")
    print(json.dumps({k: {kk: v[kk] for kk in ("device_ms", "iters_p50", "iters_max", "converged_frac", "iterations_launched")} for k, v in out.items()}))
else:
    for rep in range(2):
        for mode in ("0", "1"):
            env = dict(os.environ, OH_TQ_EVAL3=mode)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print("OH_TQ_EVAL3=" + mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
