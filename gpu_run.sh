cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line.json; python -c "
import json; j=json.load(open('gpurun_out/bench_line.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['solve_kernel'], {k:v for k,v in j['cpu_baseline'].items() if 'value' in k or 'cores' in k})"
