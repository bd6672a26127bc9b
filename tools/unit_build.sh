#!/bin/bash
# development aid: recompile ONE model unit of kernels.hip and relink (see tools/quad_build.sh).   tools/unit_build.sh unicycle CORBO_HIP_DYN_UNICYCLE [extra flags]
set -e
NAME=$1; DYN=$2; shift 2
cd "$(dirname "$0")/../control_box_rst_amd/csrc"
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -Wno-pass-failed -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc $FLAGS "-DCORBO_HIP_DYN_TU=$DYN" "-DCORBO_HIP_DYN_TU_NAME=$NAME" "$@" -c kernels.hip -o _obj/kernels_$NAME.o -Rpass-analysis=kernel-resource-usage 2> /tmp/unit.remarks || { grep -E "error" /tmp/unit.remarks | head; exit 1; }
grep -A12 "Function Name: .*hessian_kernel" /tmp/unit.remarks | grep -E "Function Name|VGPRs:|Scratch|Occupancy" | sed 's/.*remark: //' | cut -c1-120
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o libcorbo_hip.so _obj/*.o
echo "relinked libcorbo_hip.so"
