#!/bin/bash
# Headline bench under settings of a handle's algorithm options (OH_DEBUG_OPTIONS), one box, back to back.
# usage: gpurun -- 'bash tools/gpu_headline_options.sh "relax=1.3" "relax=1.7" ...'   -> gpurun_out/headline_options.txt
mkdir -p gpurun_out
out=gpurun_out/headline_options.txt
: > $out
for opt in "" "$@" ""; do
  echo "== $opt" >> $out
  OH_DEBUG_OPTIONS="$opt" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); q=d.get('quality',{}); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['launches'], q.get('converged_frac'), q.get('iters_mean'))" >> $out
done
cat $out
