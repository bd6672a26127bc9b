#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts per access pattern (tools/fetch_calibration.hip); runs on the GPU box:
#   bash tools/fetch_calibration.sh > profiles/rNN_fetch_size_calibration.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/calib; rm -rf "$OUT"; mkdir -p "$OUT"
BIN=$ROOT/tools/fetch_calibration
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/fetch_calibration.hip" -o "$BIN" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o q --output-format csv -- "$BIN" > "$OUT/f.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/w" -o q --output-format csv -- "$BIN" > "$OUT/w.log" 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
known = {}
for line in open(out + "/f.log"):
    if line.startswith("KNOWN "):
        name, b = line[6:].rsplit(" ", 1)
        known[name.strip()] = int(b)
def counters(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
def lines_touched(n, pitch, run):   # 128-byte lines a run of `run` bytes every `pitch` bytes touches, n runs
    return sum((r * pitch + run - 1) // 128 - (r * pitch) // 128 + 1 for r in range(n)) * 128
GiB = 1 << 30
lines = {"read_rec": lines_touched(GiB // (616 * 8), 616 * 8, 144 * 8), "read_row12": lines_touched(GiB // (28 * 8) // 5 * 5, 28 * 8, 12 * 8)}
print("rocprofv3 counter (KiB units x 1024) against the bytes the lanes ask for, 1 GiB buffer (4 x the Infinity Cache), mean of 3 launches per pattern")
print(f"{'pattern':52s} {'counter':>10s} {'known bytes':>14s} {'counter bytes':>14s} {'ratio':>7s} {'vs 128-B lines touched':>23s}")
for sub, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    acc = counters(sub, cn)
    for k, v in sorted(acc.items()):
        kk = k.replace("> >", ">>")
        name = next((n for n in known if kk.startswith(n) or kk.startswith("void " + n)), None)
        if name is None or (cn == "FETCH_SIZE") != name.startswith("read"):
            continue
        cb = sum(v) / len(v) * 1024.0
        short = name.split("(")[0].replace("HIP_vector_type<double, 2u>", "double2")
        ln = lines.get(short)
        print(f"{short:52s} {cn:>10s} {known[name]:14d} {cb:14.0f} {cb / known[name]:7.3f} {(f'{cb / ln:.3f} ({ln} B)' if ln else ''):>23s}")
PY
