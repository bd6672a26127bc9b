export CORBO_HIP_LIB=$PWD/control_box_rst_amd/csrc/libcorbo_hip_dev.so
python tools/opt_probe.py lag_priority=0 lag_priority=1 lag_priority=0 lag_priority=1 2>&1 | grep -v amdgpu.ids
for i in 900; do OPTS=lag_priority=1 python tools/pass_timeline.py $i 1024 2>&1 | grep "pass timeline\|priority"; done
