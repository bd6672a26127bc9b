#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 summaries behind the numbers of bench.py, the current round (CORBO_PROFILE_ROUND).  Output: gpurun_out/prof/ (scratch)
# and the files to be judged copied into profiles/ of the box's repo copy, then mirrored under gpurun_out/prof/profiles/ (copy THOSE into
# profiles/ in the build container and commit).  Counter passes are separate runs without any trace domain other than kernel-trace.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
RND=${CORBO_PROFILE_ROUND:-r06}; export CORBO_PROFILE_ROUND=$RND
TAG="${1:-round ${RND#r0}}"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || echo "rocprofv3 $name failed ($?)"; }
first() { find "$OUT/$1" -name "$2" | head -1; }
# ---- kernel traces
run bench  --kernel-trace --stats -d "$OUT/bench" -o bench --output-format csv -- python "$ROOT/bench.py" --no-cpu-baseline --no-secondary
run sweep  --kernel-trace --stats -d "$OUT/sweep" -o sweep --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 50
run cfg5   --kernel-trace --stats -d "$OUT/cfg5" -o cfg5 --output-format csv -- python "$ROOT/bench.py" --config 5 --no-cpu-baseline
run cfg5ns --kernel-trace --stats -d "$OUT/cfg5ns" -o cfg5ns --output-format csv -- python "$ROOT/tools/profile_cfg5.py" 512 3 0 nospec
run loop   --kernel-trace --stats -d "$OUT/loop" -o loop --output-format csv -- python "$ROOT/tools/mpc_loop.py" 1024 100 10 call
run band   --kernel-trace --stats -d "$OUT/band" -o band --output-format csv -- python "$ROOT/tools/band_time.py"
run freedt --kernel-trace --stats -d "$OUT/freedt" -o freedt --output-format csv -- python "$ROOT/tools/free_dt_time.py" 100
run xe     --kernel-trace --stats -d "$OUT/xe" -o xe --output-format csv -- python "$ROOT/tools/xe_time.py"
run long   --kernel-trace --stats -d "$OUT/long" -o long --output-format csv -- python "$ROOT/tools/long_horizon_time.py"
BATCH=8192 run b8192  --kernel-trace --stats -d "$OUT/b8192" -o b8192 --output-format csv -- python "$ROOT/tools/opt_probe.py" ""
# ---- HBM traffic counters (separate passes): the stand-alone sweep, and the run-to-completion solve kernel
run sfetch --pmc FETCH_SIZE -d "$OUT/sfetch" -o q --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 10
run swrite --pmc WRITE_SIZE -d "$OUT/swrite" -o q --output-format csv -- python "$ROOT/tools/profile_sweep.py" 1024 10
run lfetch --pmc FETCH_SIZE -d "$OUT/lfetch" -o q --output-format csv -- python "$ROOT/bench.py" --solve-only --steps 10 --warmup 2
run lwrite --pmc WRITE_SIZE -d "$OUT/lwrite" -o q --output-format csv -- python "$ROOT/bench.py" --solve-only --steps 10 --warmup 2
# ---- issue / wait counters of the solve kernel, fp64 matrix-core counters of the cfg-5 kernels
run sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/sq1" -o q --output-format csv -- python "$ROOT/bench.py" --solve-only --steps 5 --warmup 1
run sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d "$OUT/sq2" -o q --output-format csv -- python "$ROOT/bench.py" --solve-only --steps 5 --warmup 1
run hess   --kernel-trace --stats -d "$OUT/hess" -o hess --output-format csv -- python "$ROOT/tools/hessian_split_ab.py" 1024 0
run hess1  --kernel-trace --stats -d "$OUT/hess1" -o hess1 --output-format csv -- python "$ROOT/tools/hessian_split_ab.py" 1 2
run mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d "$OUT/mfma" -o q --output-format csv -- python "$ROOT/tools/profile_cfg5.py" 512 3
run qfetch --pmc FETCH_SIZE -d "$OUT/qfetch" -o q --output-format csv -- python "$ROOT/tools/profile_cfg5.py" 512 3
run qwrite --pmc WRITE_SIZE -d "$OUT/qwrite" -o q --output-format csv -- python "$ROOT/tools/profile_cfg5.py" 512 3
cd "$ROOT"
P=$ROOT/profiles
for n in bench sweep cfg5 cfg5ns loop band freedt xe long b8192 hess hess1; do s=$(first $n "*kernel_stats.csv"); [ -n "$s" ] && cp "$s" "$P/${RND}_${n}_kernel_stats.csv"; done; [ -f "$P/${RND}_hess_kernel_stats.csv" ] && mv "$P/${RND}_hess_kernel_stats.csv" "$P/${RND}_hessian_kernel_stats.csv"; [ -f "$P/${RND}_hess1_kernel_stats.csv" ] && mv "$P/${RND}_hess1_kernel_stats.csv" "$P/${RND}_hessian_single_kernel_stats.csv"; [ -f "$P/${RND}_cfg5ns_kernel_stats.csv" ] && mv "$P/${RND}_cfg5ns_kernel_stats.csv" "$P/${RND}_cfg5_nospec_kernel_stats.csv"
python tools/summarize_pmc.py sweep "$(first sfetch '*counter_collection.csv')" "$(first swrite '*counter_collection.csv')" 1024 100 "$TAG, tools/profile_sweep.py 1024 10" | cut -c1-300
python tools/summarize_pmc.py solve "$(first lfetch '*counter_collection.csv')" "$(first lwrite '*counter_collection.csv')" 1024 100 "$TAG, bench.py --solve-only --steps 10 --warmup 2" | cut -c1-300
python tools/summarize_pmc.py sq "$(first sq1 '*counter_collection.csv')" "$(first sq2 '*counter_collection.csv')" 1024 100 "$TAG, bench.py --solve-only --steps 5 --warmup 1" | cut -c1-600
python tools/summarize_pmc.py cfg5 "$(first qfetch '*counter_collection.csv')" "$(first qwrite '*counter_collection.csv')" 512 200 "$TAG, tools/profile_cfg5.py 512 3" | cut -c1-900
python tools/summarize_pmc.py mfma "$(first mfma '*counter_collection.csv')" 512 200 "$TAG, tools/profile_cfg5.py 512 3" | cut -c1-900
cp "$(first sfetch '*counter_collection.csv')" "$P/${RND}_sweep_pmc_fetch_counter_collection.csv"; cp "$(first swrite '*counter_collection.csv')" "$P/${RND}_sweep_pmc_write_counter_collection.csv"
# ---- per-pass phase table of the slowest headline instance, the device's deviation per ledger fixture
python tools/pass_timeline.py 900 > "$P/${RND}_pass_phases_instance900.txt" 2>&1
python tools/ledger_check.py > "$P/${RND}_ledger_device_deviation.txt" 2>&1
bash tools/fetch_calibration.sh > "$P/${RND}_fetch_size_calibration.txt" 2>&1   # FETCH_SIZE / WRITE_SIZE against known byte counts per access pattern
cd "$ROOT"
{ python tools/free_dt_time.py 100; python tools/pquad_time.py; } 2>&1 | grep -v amdgpu.ids > "$P/${RND}_free_dt_routes.txt"
{ python tools/xe_time.py; python tools/xe_batch_sweep.py; python tools/narrow_band_phases.py; } 2>&1 | grep -v amdgpu.ids > "$P/${RND}_extra_edge_routes.txt"
{ python tools/bt_phases.py 0 1; python tools/bt_phases.py 900 1024; } 2>&1 | grep -v amdgpu.ids | cut -c1-400 > "$P/${RND}_xe_pass_phases.txt"
# ---- the bench lines of the same build on the same box (the profile-derived fields now resolve against the files above)
python bench.py > "$P/${RND}_bench_line.json" 2> "$OUT/benchline3.err"
python bench.py --config 5 > "$P/${RND}_bench_line_cfg5.json" 2> "$OUT/benchline5.err"
python bench.py --config 2 > "$P/${RND}_bench_line_cfg2.json" 2> "$OUT/benchline2.err"
python bench.py --config 1 > "$P/${RND}_bench_line_cfg1.json" 2> "$OUT/benchline1.err"
mkdir -p "$OUT/profiles"; cp "$P"/${RND}_* "$P"/sweep_pmc_latest.json "$OUT/profiles/"
head -3 "$P/${RND}_bench_kernel_stats.csv" | cut -c1-200; head -3 "$P/${RND}_sweep_kernel_stats.csv" | cut -c1-200; head -5 "$P/${RND}_cfg5_kernel_stats.csv" | cut -c1-200
for c in "" _cfg5 _cfg2 _cfg1; do python -c "
import json,sys
j=json.loads(open('$P/${RND}_bench_line$c.json').read().strip().splitlines()[-1])
print('$c', j['value'], j['ms_per_step'], j.get('roofline',{}).get('frac'), j.get('cpu_baseline',{}).get('value'))"; done
