cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for n in 1 2 4; do
  CORBO_HIP_SUBBATCHES=$n python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nsub=$n', {k:d[k] for k in ['value','ms_per_step']})"
done
