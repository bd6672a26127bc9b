"""Summarise rocprofv3 --pmc counter CSVs of tools/profile_sweep.py into profiles/sweep_pmc_latest.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB-units per dispatch.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
under-reports wide coalesced streaming reads by exactly 2x; this kernel's reads are 16-byte coalesced vertex loads plus
8-byte bound loads, so both the raw and the doubled figure are recorded (WRITE_SIZE is uncalibrated there: recorded raw).

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> <N> <tag>
"""
import csv
import json
import os
import sys


def per_dispatch(path, counter, kernel_substr):
    vals = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and kernel_substr in r.get("Kernel_Name", ""):
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, batch, N, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    f = per_dispatch(fetch_csv, "FETCH_SIZE", "sweep_kernel")
    w = per_dispatch(write_csv, "WRITE_SIZE", "sweep_kernel")
    f_avg, w_avg = sum(f) / len(f), sum(w) / len(w)
    out = {
        "batch": batch, "N": N, "tag": tag, "dispatches": len(f),
        "FETCH_SIZE_KiB_per_launch_raw": f_avg, "WRITE_SIZE_KiB_per_launch_raw": w_avg,
        "hbm_bytes_per_launch_raw": (f_avg + w_avg) * 1024.0,
        "hbm_bytes_per_launch": (2.0 * f_avg + w_avg) * 1024.0,
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/profile_sweep.py, {tag}; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction, WRITE_SIZE raw",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "sweep_pmc_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
