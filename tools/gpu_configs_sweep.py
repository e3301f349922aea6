"""Round 6 (verdict Next 3a): where does each of the families behind BASELINE configs 1, 3, 4, 5 saturate the machine?  Batch sweeps of the GPU legs of
tools/bench_configs.py (same instance generators, no oracle, no CPU leg): device ms of one batched solve, solves/s, and the roofline object of the
dominant kernel group at every size (f64 flop per work unit from profiles/configs_flops.json x the run's work / its HIP-event time).

  python tools/gpu_configs_sweep.py [families...] > gpurun_out/profiles/r06_configs_sweep.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402

SWEEPS = {
    "config1_ik": [4096, 16384, 65536, 262144, 1048576],
    "config3_point_mass": [4096, 16384, 65536, 262144, 1048576],
    "config4_arms_r0.15": [256, 1024, 4096, 16384, 65536],
    "config5_torque": [1024, 4096, 8192, 16384, 32768, 65536],
}


def row(v):
    rf = v.get("roofline") or {}
    return {"batch": v["batch"], "device_ms": v["device_ms"], "solves_per_s": v["solves_per_s"], "converged_frac": v["converged_frac"], "iters_p50": v["iters_p50"],
            "iters_max": v["iters_max"], "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "frac", "kernel_ms", "work_units", "all_kernels")}}


def main():
    fams = sys.argv[1:] or list(SWEEPS)
    out = {"what": __doc__.split("\n\n")[0], "peak_f64_tflops": bc.F64_PEAK_TFLOPS}
    for fam in fams:
        rows = []
        for B in SWEEPS[fam]:
            t0 = time.perf_counter()
            try:
                if fam == "config1_ik":
                    v = bc.run_configs(sample=0, cpu=False, only="ik", ik_batch=B)["config1_ik"]
                elif fam == "config3_point_mass":
                    v = bc.run_configs(sample=0, cpu=False, only="pm", pm_batch=B)["config3_point_mass"]
                elif fam == "config4_arms_r0.15":
                    v = bc.run_configs(sample=0, cpu=False, only="config4", config4_cases=((B, 0.15),))[f"config4_arms{B}_r0.15"]
                else:
                    v = bc.run_configs(sample=0, cpu=False, only="torque", torque_batches=(B,))[f"config5_torque_b{B}"]
                rows.append({**row(v), "wall_s": time.perf_counter() - t0})
            except Exception as e:  # noqa: BLE001
                rows.append({"batch": B, "error": f"{type(e).__name__}: {e}"})
                break
            print(fam, json.dumps(rows[-1])[:300], file=sys.stderr, flush=True)
        ok = [r for r in rows if "solves_per_s" in r]
        best = max(ok, key=lambda r: r["solves_per_s"]) if ok else None
        out[fam] = {"rows": rows, "saturates_at": None if best is None else {"batch": best["batch"], "solves_per_s": best["solves_per_s"]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
