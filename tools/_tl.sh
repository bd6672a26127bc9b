for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --preheat-ms 0"; do
python3 bench.py --gpus 1 $a --solve-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$a', round(j['ms_per_step'],4), j['preheat']['untimed_steps'])"
done
