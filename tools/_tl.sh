timeout 1500 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -5
