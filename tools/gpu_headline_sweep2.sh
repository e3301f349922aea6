#!/bin/bash
# headline on two streams: compaction threshold and hand-over threshold re-swept (they were tuned on one stream); one box, back to back
for opts in "" "compact_frac=0.95" "compact_frac=0.93" "compact_frac=0.90" "tail_threshold=8192" "tail_threshold=32768" "tail_threshold=24576" "check_every=2" ""; do
  v=$(OH_DEBUG_OPTIONS="$opts" python bench.py --steps 3 --warmup 1 --timed-only --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d.get('compactions_per_step'), d.get('tail_iteration_frac'))")
  echo "[$opts] $v"
done
