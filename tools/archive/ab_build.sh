#!/bin/bash
# Build variants of liboptas_hip.so for A/B runs on ONE GPU box (box-to-box variance is ~8 %):
#   tools/ab_build.sh name1 "-DFLAG=1" name2 "-DFLAG=2" ...   ->  build_abl/lib_<name>.so
# then on the box:  for v in name1 name2; do cp build_abl/lib_$v.so optas_amd/liboptas_hip.so; python bench.py ...; done
set -e
mkdir -p build_abl
python -c "from optas_amd.build import embed_solver_source, embed_jit_headers; embed_solver_source(); embed_jit_headers()"
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -shared -fPIC -Iinclude -Ioptas_amd/csrc $flags -o build_abl/lib_$name.so \
    optas_amd/csrc/oh_kernels.hip optas_amd/csrc/oh_jit.hip optas_amd/csrc/oh_fkjac.hip optas_amd/csrc/oh_torque.hip optas_amd/csrc/oh_free.hip optas_amd/csrc/oh_pointmass.hip optas_amd/csrc/oh_ik.hip optas_amd/csrc/oh_qp.hip \
    optas_amd/csrc/oh_tape.hip optas_amd/csrc/oh_api.hip -lhiprtc &
done
wait
ls -la build_abl
