#!/bin/bash
mkdir -p gpurun_out/r6m gpurun_out/profiles
bash tools/gpu_full.sh r6m > gpurun_out/r6m/full.log 2>&1
OH_DEBUG_OPTIONS=streams=1 timeout 1200 bash tools/profile.sh r06 "--steps 5 --warmup 2 --no-cpu-baseline --timed-only" > gpurun_out/r6m/profile.log 2>&1
tail -30 gpurun_out/r6m/full.log | cut -c1-300
head -12 gpurun_out/profiles/r06_kernel_stats.csv
