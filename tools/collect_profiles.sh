#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 summaries behind the numbers of bench.py.  Output: gpurun_out/prof/*.csv|json
# (copy what should be judged into profiles/).  Counter passes are separate runs without any trace domain other than kernel-trace.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || echo "rocprofv3 $name failed ($?)"; }
run bench  --kernel-trace --stats -d "$OUT/bench" -o bench --output-format csv -- python "$ROOT/bench.py" --no-cpu-baseline
run loop   --kernel-trace --stats -d "$OUT/loop" -o loop --output-format csv -- python "$ROOT/tools/mpc_loop.py" 1024 100 10 call
run sweep  --kernel-trace --stats -d "$OUT/sweep" -o sweep --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 50
run fetch  --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 10
run write  --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 10
find "$OUT" -name "*.csv" | sed "s|$OUT/||"
f=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1); w=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
cd "$ROOT" && python tools/summarize_pmc.py "$f" "$w" 1024 100 "${1:-round 1}" && cp profiles/sweep_pmc_latest.json "$OUT/"
for n in bench sweep loop; do s=$(find "$OUT/$n" -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" "$OUT/${n}_kernel_stats.csv"; done
cp "$f" "$OUT/sweep_pmc_fetch_counter_collection.csv"; cp "$w" "$OUT/sweep_pmc_write_counter_collection.csv"
grep -h "sweep\|factor" "$OUT"/sweep.log | tail -3
head -5 "$OUT/bench_kernel_stats.csv"; head -4 "$OUT/sweep_kernel_stats.csv"
tail -1 "$OUT/bench.log" | cut -c1-300
