cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
python tools/profile_sweep.py 1024 50
python tools/profile_sweep.py 1 50
python tools/profile_sweep.py 4096 20
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','batch_steps_per_s','kernel_split_ms','roofline']})"
