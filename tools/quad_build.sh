#!/bin/bash
# development aid: recompile ONLY the quadrotor unit of kernels.hip (big-block family) and relink libcorbo_hip.so with the other objects of
# the last full build (valid while the edit touches nothing the other units instantiate).   tools/quad_build.sh [extra hipcc flags...]
set -e
cd "$(dirname "$0")/../control_box_rst_amd/csrc"
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -Wno-pass-failed -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc $FLAGS -DCORBO_HIP_DYN_TU=CORBO_HIP_DYN_QUADROTOR -DCORBO_HIP_DYN_TU_NAME=quadrotor -DCORBO_HIP_DYN_TU_BIG "$@" \
    -c kernels.hip -o _obj/kernels_quadrotor.o -Rpass-analysis=kernel-resource-usage 2> /tmp/quad.remarks || { grep -E "error|warning: v" /tmp/quad.remarks | head -30; exit 1; }
for k in big_chain3 big_chain2 big_stage; do
  grep -A12 "Function Name: .*$k" /tmp/quad.remarks | grep -E "Function Name|VGPRs:|AGPRs|Scratch|Occupancy|LDS Size" | sed 's/.*remark: //' | tr '\n' ' '; echo
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o libcorbo_hip.so _obj/*.o
echo "relinked libcorbo_hip.so"
