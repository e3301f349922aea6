#!/bin/bash
# headline with / without the progress sort of the compaction, one and two streams, one box, back to back (value = solves/s)
for opts in "" "compact_sort=0" "streams=1" "streams=1,compact_sort=0" ""; do
  v=$(OH_DEBUG_OPTIONS="$opts" python bench.py --steps 3 --warmup 1 --timed-only --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d.get('compactions_per_step'))")
  echo "[$opts] $v"
done
