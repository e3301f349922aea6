#!/bin/bash
mkdir -p gpurun_out/r6d
timeout 900 bash tools/gpu_configs_pmc.sh > gpurun_out/r6d/pmc.log 2>&1
cp gpurun_out/profiles/r06_configs_flops.json profiles/configs_flops.json
timeout 300 python -m pytest tests/test_gpu_options.py -m gpu -q -x > gpurun_out/r6d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6d/pytest.log
( time python bench.py > gpurun_out/r6d/bench.json 2> gpurun_out/r6d/bench.err ) 2> gpurun_out/r6d/bench.time
tail -3 gpurun_out/r6d/pytest.log; cat gpurun_out/r6d/bench.time; tail -5 gpurun_out/r6d/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6d/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"])
c=j["configs"]
for k,v in c.items():
    if isinstance(v, dict):
        rf=v.get("roofline") or {}; cb=v.get("cpu_baseline") or {}
        print(k, v.get("device_ms"), "| roofline", rf.get("kernel"), rf.get("achieved"), rf.get("frac"), rf.get("kernel_ms"), "| cpu", cb.get("value"), cb.get("cores"), cb.get("gpu_vs_port_objective_rel_max"), cb.get("error"))
    else: print(k, str(v)[:300])
PY
