#!/bin/bash
# bench.py over batch sizes on one box: one JSON line per size (profiles/rNN_batch_sweep.json)
for B in ${SWEEP_BATCHES:-1024 4096 8192 16384 32768 65536 131072 262144 393216}; do
  python bench.py --batch $B --steps 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
dom = max(r['all_kernels'].items(), key=lambda kv: kv[1]['total_ms']) if r.get('all_kernels') else (r.get('kernel'), {})
print(json.dumps({'batch': $B, 'solves_per_s': round(d['value']), 'ms_per_batch': round(d['ms_per_step'], 3), 'roofline_frac_k_eval': round(r['frac'], 3) if r.get('frac') else None,
                  'dominant': dom[0], 'iters_p50': d['quality']['iters_p50'], 'converged': d['quality']['converged_frac'], 'tail_iteration_frac': d.get('tail_iteration_frac')}))"
done
