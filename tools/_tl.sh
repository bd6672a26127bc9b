timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hessian.py tests/test_gpu_dropin.py -m gpu -q -x -k "kcar or dropin or reference_ocp" 2>&1 | tail -8
python tools/opt_probe.py lag_priority=1 2>&1 | grep -v amdgpu.ids
