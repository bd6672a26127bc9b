cd $GRAFT_REPO_ROOT
tools/collect_profiles.sh "round 1, component-centric sweep (one prefetched descriptor per component, LDS staging, 4 workgroups / CU)"
python bench.py 2>/dev/null | tail -1 > gpurun_out/prof/bench_line.json
