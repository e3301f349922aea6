"""config 4 synthetic only (device time, iterations, oracle-graded sample): python tools/gpu_cfg4_only.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import run_configs  # noqa: E402

o = run_configs(sample=8, only="config4")
for k, v in o.items():
    print(k, json.dumps({kk: vv for kk, vv in v.items() if kk != "what"}))
