#!/bin/bash
# round 6, call b: GPU suite with the deferred refactorisation, A/B against the same sources with the in-kernel retry loop, fork rate at 1e-8
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/pytest.log
bash tools/gpu_ab_lib.sh optas_amd/liboptas_hip_nodefer.so > gpurun_out/r6b/ab_defer.log 2>&1
python tools/gpu_fork_rate.py 262144 1e-8 _r6b > gpurun_out/r6b/fork8.log 2>&1
tail -3 gpurun_out/r6b/pytest.log; cat gpurun_out/r6b/ab_defer.log
