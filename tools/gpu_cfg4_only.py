import sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
from bench_configs import run_configs
o=run_configs(sample=0, only='config4')
print({k:(round(v['device_ms'],2), v['iters_max'], v['converged_frac']) for k,v in o.items() if isinstance(v,dict)})
