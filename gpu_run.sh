cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k horizon 2>&1 | grep -E "FAILED|passed|failed|Error|assert " | head -30
