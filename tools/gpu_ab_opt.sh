#!/bin/bash
# A/B of one option on one box, interleaved: tools/gpu_ab_opt.sh "retract_xcd=0" "retract_xcd=1" [common options]
A=$1; B=$2; C=${3:-}
run() { OH_DEBUG_OPTIONS="$1${C:+,$C}" python bench.py --steps 3 --warmup 1 --timed-only --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels']; print(round(d['value']), round(d['ms_per_step'],2), 'k_eval', round(k['k_eval']['avg_launch_ms'],4), 'k_step', round(k['k_step']['avg_launch_ms'],4), d['quality']['iters_p50'], d['quality']['iters_max'], d['quality']['converged_frac'])"; }
for i in 1 2 3; do echo "[$A] $(run $A)"; echo "[$B] $(run $B)"; done
