# diagnostics: per-pass phase log of one instance, 256- vs 128-thread workgroups (development library if present)
LIB=control_box_rst_amd/csrc/libcorbo_hip_tl.so; [ -f $LIB ] && export CORBO_HIP_LIB=$PWD/$LIB
for T in ${THREADS_LIST:-256 128}; do
 for spec in "0 1" "900 1024"; do
  echo "== threads $T inst/batch $spec"; OPTS=pass_threads=$T python tools/pass_timeline.py $spec 2>&1 | grep -v amdgpu.ids | cut -c1-400
 done
done
