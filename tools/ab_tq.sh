for v in tq_base tq_e2s2 tq_e3s2 tq_base; do cp build_abl/lib_$v.so optas_amd/liboptas_hip.so; echo $v; TQ_CASES=100:300 python tools/gpu_torque_b8192.py 2>&1 | tail -1; done
