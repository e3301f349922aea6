#!/bin/bash
# on the GPU box: bench each variant built by tools/ab_build.sh (same box, back to back)
for v in "$@"; do
  cp build_abl/lib_$v.so optas_amd/liboptas_hip.so
  python bench.py --no-cpu-baseline --no-configs --steps 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['quality']['iters_p50'], d['quality']['f_mean'], 'per-launch us', {k: round(v['avg_launch_ms']*1e3,1) for k,v in d['roofline']['all_kernels'].items()}, 'units/launch', round(d['roofline']['units_per_launch_avg']), 'fk frac', round(d['roofline_fk_jac']['frac'],3), round(d['roofline_fk_jac']['avg_launch_ms'],3))"
done
