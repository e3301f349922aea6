#!/bin/bash
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -m gpu -q -k "not bench_scale" > gpurun_out/r6c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6c/pytest.log
python bench.py > gpurun_out/r6c/bench.json 2> gpurun_out/r6c/bench.err
tail -4 gpurun_out/r6c/pytest.log; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6c/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"])
for k in ("tol_1e-08","tol_1e-06","batch_invariant","pcie_inclusive"): print(k, json.dumps(j.get(k))[:600])
print(json.dumps(j["quality"].get("population_vs_host_port"))[:400])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
PY
