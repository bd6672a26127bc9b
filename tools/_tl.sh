python tools/time_hessian_ops.py 1024 10 2>&1 | grep -v amdgpu
python tools/time_hessian_ops.py 1 200 2>&1 | grep -v amdgpu | head -1
timeout 900 python -m pytest tests/test_gpu_hessian.py -m gpu -q -x -n 4 2>&1 | tail -2
