#!/bin/bash
mkdir -p gpurun_out/r6g gpurun_out/profiles
timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_bench_scale.py -m gpu -q -x > gpurun_out/r6g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6g/pytest.log
OH_DEBUG_OPTIONS=streams=1 timeout 1200 bash tools/profile.sh r06 "--steps 5 --warmup 2 --no-cpu-baseline --timed-only" > gpurun_out/r6g/profile.log 2>&1
timeout 900 bash tools/gpu_configs_pmc.sh > gpurun_out/r6g/pmc.log 2>&1
cp gpurun_out/profiles/r06_configs_flops.json profiles/configs_flops.json
timeout 1200 python tools/gpu_configs_sweep.py > gpurun_out/profiles/r06_configs_sweep.json 2> gpurun_out/r6g/sweep.err
python bench.py --no-configs > gpurun_out/r6g/bench.json 2> gpurun_out/r6g/bench.err
tail -3 gpurun_out/r6g/pytest.log
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6g/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], json.dumps(j["pcie_inclusive"])[:900])
PY
