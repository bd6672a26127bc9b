# diagnostics: headline solve under handle options with a development library.  tools/ab_probe.sh <lib suffix> <opt_probe args...>
LIB=control_box_rst_amd/csrc/libcorbo_hip_$1.so; shift   # (suffix "none": the product library)
[ -f $LIB ] && export CORBO_HIP_LIB=$PWD/$LIB
python tools/opt_probe.py "$@" 2>&1 | grep -v amdgpu.ids
