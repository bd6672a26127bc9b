cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01.json; cut -c1-400 gpurun_out/bench_r01.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01 -o r01_bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r01.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01s -o r01_sweep -- python $R/tools/profile_sweep.py 1024 50 > $R/gpurun_out/prof_r01s.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/profile_sweep.py 1024 5 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/profile_sweep.py 1024 5 > $R/gpurun_out/pmc_write.log 2>&1
cat $R/gpurun_out/prof_r01/r01_bench_kernel_stats.csv; cat $R/gpurun_out/prof_r01s/r01_sweep_kernel_stats.csv; cat $R/gpurun_out/prof_r01s.log | tail -2
