cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/profile_sweep.py 1024 20
python tools/profile_sweep.py 1 20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/tools/profile_sweep.py 1024 5 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/tools/profile_sweep.py 1024 5 > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write; head -3 $R/gpurun_out/pmc_fetch/*counter_collection.csv
cd $R && python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/bench2.json; cat gpurun_out/bench2.json | cut -c1-1500
