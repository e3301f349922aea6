#!/bin/bash
# the driver's round-end sequence on one box: GPU suite, smoke(), the bench line (-> gpurun_out/<tag>/)
TAG=${1:-full}; mkdir -p gpurun_out/$TAG
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.log 2>&1
( time python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ) 2> gpurun_out/$TAG/bench.time
tail -4 gpurun_out/$TAG/pytest.log; tail -1 gpurun_out/$TAG/smoke.log; cat gpurun_out/$TAG/bench.time | tail -3
python - <<PY
import json
j=json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "fk", j["roofline_fk_jac"]["frac"])
for k in ("tol_1e-08","tol_1e-06","batch_invariant"): print(k, {kk: vv for kk, vv in (j.get(k) or {}).items() if kk != "note"})
print("pcie", j["pcie_inclusive"]["solves_per_s"], "pop", j["quality"].get("population_vs_host_port",{}).get("misses_1e-6"), j["quality"].get("population_vs_host_port",{}).get("misses_1e-9"))
print("lat", j["latency_b1_ms"], j["latency_b1024_ms"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
for k,v in j["configs"].items():
    if isinstance(v, dict):
        rf=v.get("roofline") or {}; cb=v.get("cpu_baseline") or {}
        print(k, v.get("device_ms"), v.get("converged_frac"), "| rf", rf.get("kernel"), rf.get("achieved"), rf.get("frac"), "| cpu", cb.get("value"), cb.get("cores"), cb.get("gpu_vs_port_objective_rel_max"), cb.get("error"))
PY
