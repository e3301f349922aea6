#!/bin/bash
# on the GPU box: the bench under several settings of one environment variable, interleaved and repeated (run-to-run variance of the
# memory-bound kernels is ~7 %: the placement of the pool differs from process to process)
#   tools/ab_env.sh VAR "v1 v2 ..." repeats
VAR=$1; VALS=$2; REP=${3:-3}
for r in $(seq $REP); do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-configs --steps 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), round(d['ms_per_step'],1), {k: round(x['avg_launch_ms']*1e3,1) for k,x in d['roofline']['all_kernels'].items()}, round(d['roofline_fk_jac']['frac'],3))"
done; done
