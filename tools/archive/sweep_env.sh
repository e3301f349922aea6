#!/bin/bash
# A/B on one box: bench.py under different values of one environment variable.  usage: tools/sweep_env.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --steps 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'])"
done
