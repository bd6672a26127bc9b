#!/bin/bash
# Attribution of the solve kernel's write traffic (VERDICT r2 "what's weak" 2): WRITE_SIZE / FETCH_SIZE of the shipping build (128 VGPRs, 44 spilled) against a
# 168-VGPR build of the same source (-DCORBO_HIP_PASS_WAVES=3: 4 spilled registers, three workgroups per CU) -- tools/dev_build.sh nospill -DCORBO_HIP_PASS_WAVES=3
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/spill
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in ship nospill; do
  if [ $lib = nospill ]; then export CORBO_HIP_LIB=$ROOT/control_box_rst_amd/csrc/libcorbo_hip_nospill.so; else unset CORBO_HIP_LIB; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --pmc $c -d "$OUT/${lib}_$c" -o q --output-format csv -- python "$ROOT/bench.py" --solve-only --steps 10 --warmup 2 --no-secondary > "$OUT/${lib}_$c.log" 2>&1
  done
done
python - <<PY
import csv, glob, json, os
out = {}
for lib in ("ship", "nospill"):
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (lib, c), recursive=True)
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f[0])):
            if "lm_pass_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
        # per dispatch the counter is reported once per XCD-instance row set; n rows = dispatches x instances
        disp = len({r["Dispatch_Id"] for r in csv.DictReader(open(f[0])) if "lm_pass_kernel" in r["Kernel_Name"]})
        out["%s_%s_raw_per_launch" % (lib, c)] = tot / max(1, disp)
        out["%s_launches" % lib] = disp
print(json.dumps(out))
json.dump(out, open("$OUT/spill_attribution_raw.json", "w"), indent=1)
PY
