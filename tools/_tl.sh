timeout 900 python -m pytest tests/test_gpu_multigpu_cpp.py -m gpu -q -x 2>&1 | tail -6
examples/multi_gpu_batch 2 2048 100
examples/multi_gpu_batch 1 1024 100
