#!/bin/bash
# development aid: compile ONLY the unicycle unit of kernels.hip (headline defect formula) with extra flags and link it with the
# other objects of the last full build into csrc/libcorbo_hip_dev.so (A/B against the product library with CORBO_HIP_LIB=...)
#   tools/dev_build.sh [name] [extra hipcc flags...]      -> control_box_rst_amd/csrc/libcorbo_hip_<name>.so + /tmp/<name>.s
set -e
cd "$(dirname "$0")/../control_box_rst_amd/csrc"
name=${1:-dev}; shift || true
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc $FLAGS -DCORBO_HIP_DEV_FAST -DCORBO_HIP_DYN_TU=CORBO_HIP_DYN_UNICYCLE -DCORBO_HIP_DYN_TU_NAME=unicycle "$@" \
    -c kernels.hip -o /tmp/kernels_unicycle_$name.o -Rpass-analysis=kernel-resource-usage 2> /tmp/$name.remarks || { tail -30 /tmp/$name.remarks; exit 1; }
grep -A9 "lm_pass_kernelILi2ELi3ELb0ELb1ELi101" /tmp/$name.remarks | grep -E "VGPRs|Scratch|SGPRs" | sed 's/.*remark: //' | tr '\n' ' '; echo
objs=$(ls _obj/*.o | grep -v kernels_unicycle.o)
if [ -n "$MAIN" ]; then   # MAIN=1: also the main unit (model-independent kernels: factor / big-block family / helpers) with the extra flags
    /opt/rocm/bin/hipcc $FLAGS "$@" -c kernels.hip -o /tmp/kernels_main_$name.o
    /opt/rocm/bin/hipcc $FLAGS -DCORBO_HIP_DYN_TU=CORBO_HIP_DYN_QUADROTOR -DCORBO_HIP_DYN_TU_NAME=quadrotor -DCORBO_HIP_DYN_TU_BIG "$@" -c kernels.hip -o /tmp/kernels_quad_$name.o
    objs=$(echo "$objs" | tr ' ' '\n' | grep -v "_obj/kernels.o" | grep -v "_obj/kernels_quadrotor.o"); objs="$objs /tmp/kernels_main_$name.o /tmp/kernels_quad_$name.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o libcorbo_hip_$name.so $objs /tmp/kernels_unicycle_$name.o
echo "built libcorbo_hip_$name.so"
