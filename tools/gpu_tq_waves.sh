#!/bin/bash
# Occupancy variants of k_tq_eval3 / k_tq_curv (launch bounds OH_TQ_EVAL3_WAVES / OH_TQ_CURV_WAVES), built into build_var/ by optas_amd.build.build(out=, defines=).
mkdir -p gpurun_out
: > gpurun_out/tq_waves.jsonl
python tools/gpu_tq_time.py 8192 1024 >> gpurun_out/tq_waves.jsonl
for f in build_var/liboptas_hip_*.so; do
  OPTAS_HIP_LIBRARY=$PWD/$f python tools/gpu_tq_time.py 8192 1024 >> gpurun_out/tq_waves.jsonl
done
cat gpurun_out/tq_waves.jsonl
