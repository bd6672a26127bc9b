for w in 5 100 5 300; do python bench.py --steps 20 --warmup $w --no-cpu-baseline --solve-only 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('warmup $w ms_per_step', j['ms_per_step'], j['value'])"; done
