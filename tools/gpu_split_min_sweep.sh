for cfg in "65536 streams=1" "65536 split_min=32768" "65536 streams=1" "65536 split_min=32768" "32768 streams=1" "32768 split_min=16384" "131072 streams=1" "131072 streams=2" "131072 streams=3,split_min=65536"; do
  set -- $cfg
  v=$(OH_DEBUG_OPTIONS="$2" python bench.py --batch $1 --steps 5 --warmup 1 --timed-only --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('streams_per_gpu'))")
  echo "[$cfg] $v"
done
