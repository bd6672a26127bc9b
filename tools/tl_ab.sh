# diagnostics: per-pass phase log of one instance of the headline workload (optionally with a development library: LIBSFX=<suffix>)
[ -n "$LIBSFX" ] && export CORBO_HIP_LIB=$PWD/control_box_rst_amd/csrc/libcorbo_hip_$LIBSFX.so
for T in ${THREADS_LIST:-256 128}; do
 for spec in "0 1" "900 1024"; do
  echo "== threads $T inst/batch $spec"; OPTS=pass_threads=$T python tools/pass_timeline.py $spec 2>&1 | grep -v amdgpu.ids | cut -c1-400
 done
done
