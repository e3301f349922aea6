#!/bin/bash
mkdir -p gpurun_out/r6f gpurun_out/profiles
timeout 900 bash tools/gpu_configs_pmc.sh > gpurun_out/r6f/pmc.log 2>&1
cp gpurun_out/profiles/r06_configs_flops.json profiles/configs_flops.json
timeout 1200 python tools/gpu_configs_sweep.py > gpurun_out/profiles/r06_configs_sweep.json 2> gpurun_out/r6f/sweep.err
timeout 1200 bash tools/profile.sh r06 > gpurun_out/r6f/profile.log 2>&1
tail -5 gpurun_out/r6f/profile.log; grep -c batch gpurun_out/profiles/r06_configs_sweep.json
