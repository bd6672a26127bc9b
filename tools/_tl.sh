timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_fuzz.py -m gpu -q -x -n 4 2>&1 | tail -5
python tools/opt_probe.py lag_priority=1 2>&1 | grep -v amdgpu.ids
