#!/bin/bash
# development aid: recompile ONLY the main unit of kernels.hip (host launchers, model-independent kernels) and corbo_hip.hip, then relink
# libcorbo_hip.so with the model units of the last full build (valid while the edit touches nothing those units instantiate).
set -e
cd "$(dirname "$0")/../control_box_rst_amd/csrc"
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -Wno-pass-failed -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc $FLAGS -c kernels.hip -o _obj/kernels.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -Wno-pass-failed -c corbo_hip.hip -o _obj/corbo_hip.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o libcorbo_hip.so _obj/*.o
echo "relinked libcorbo_hip.so"
