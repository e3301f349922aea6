#!/bin/bash
# A/B of two builds of the library on one box, interleaved: tools/gpu_ab_lib.sh optas_amd/liboptas_hip_old.so [OH_DEBUG_OPTIONS]
OLD=$(pwd)/$1; OPTS=${2:-}
run() { OH_DEBUG_OPTIONS="$OPTS" OPTAS_HIP_LIBRARY=$1 python bench.py --steps 3 --warmup 1 --timed-only --no-configs --no-cpu-baseline ${BENCH_EXTRA:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels']; print(round(d['value']), round(d['ms_per_step'],2), 'k_eval', round(k['k_eval']['avg_launch_ms'],4), 'k_step', round(k['k_step']['avg_launch_ms'],4), d['quality']['iters_p50'], d['quality']['iters_max'], d['quality']['converged_frac'])"; }
for i in 1 2 3; do echo "old: $(run $OLD)"; echo "new: $(run '')"; done
