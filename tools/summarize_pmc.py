"""Summarise rocprofv3 --pmc counter CSVs into the JSON files bench.py reads from profiles/.

FETCH_SIZE / WRITE_SIZE are reported in KiB-units per dispatch.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
under-reports wide coalesced streaming reads by exactly 2x; both the raw and the doubled figure are recorded (WRITE_SIZE is
uncalibrated there: recorded raw).

    python tools/summarize_pmc.py sweep <fetch.csv> <write.csv> <batch> <N> <tag>      -> profiles/sweep_pmc_latest.json
    python tools/summarize_pmc.py solve <fetch.csv> <write.csv> <batch> <N> <tag>      -> profiles/<round>_solve_pmc.json
    python tools/summarize_pmc.py mfma  <counters.csv> <batch> <N> <tag>               -> profiles/<round>_cfg5_mfma.json
    python tools/summarize_pmc.py cfg5  <fetch.csv> <write.csv> <batch> <N> <tag>      -> profiles/<round>_cfg5_pmc.json
    python tools/summarize_pmc.py sq    <counters.csv> [<counters2.csv>] <batch> <N> <tag> -> profiles/<round>_solve_sq.json
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("CORBO_PROFILE_ROUND", "r05")   # file-name prefix of the round's summaries (<round> above)


def per_dispatch(path, kernel_substr):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel_substr in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


def traffic(kind, fetch_csv, write_csv, batch, N, tag):
    kernel = "sweep_kernel" if kind == "sweep" else "lm_pass_kernel"
    f = per_dispatch(fetch_csv, kernel)["FETCH_SIZE"]
    w = per_dispatch(write_csv, kernel)["WRITE_SIZE"]
    if kind == "solve":   # full solves only (the profiled run has warm-up and timed solves of the same size; drop tiny dispatches)
        f = [v for v in f if v > 0.5 * max(f)]
        w = [v for v in w if v > 0.5 * max(w)]
    f_avg, w_avg = mean(f), mean(w)
    out = {
        "batch": batch, "N": N, "tag": tag, "kernel": kernel, "dispatches": len(f),
        "FETCH_SIZE_KiB_per_launch_raw": f_avg, "WRITE_SIZE_KiB_per_launch_raw": w_avg,
        "hbm_bytes_per_launch_raw": (f_avg + w_avg) * 1024.0,
        "hbm_bytes_per_launch": (2.0 * f_avg + w_avg) * 1024.0,
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {tag}; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 "
                  "correction, WRITE_SIZE raw",
    }
    name = "sweep_pmc_latest.json" if kind == "sweep" else RND + "_solve_pmc.json"
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print(json.dumps(out))


def cfg5_traffic(fetch_csv, write_csv, batch, N, tag):
    """HBM-side traffic of the two kernels of a cfg-5 factorisation, per full launch (the tail passes' small launches dropped)."""
    out = {"batch": batch, "N": N, "tag": tag, "kernels": {}}
    tot_raw = tot = 0.0
    for kernel in ("big_stage_kernel", "big_chain3_kernel"):
        f = per_dispatch(fetch_csv, kernel)["FETCH_SIZE"]
        w = per_dispatch(write_csv, kernel)["WRITE_SIZE"]
        f = [v for v in f if v > 0.5 * max(f)]
        w = [v for v in w if v > 0.5 * max(w)]
        fa, wa = mean(f), mean(w)
        out["kernels"][kernel] = {"dispatches_averaged": [len(f), len(w)], "FETCH_SIZE_KiB_per_launch_raw": fa, "WRITE_SIZE_KiB_per_launch_raw": wa,
                                  "hbm_bytes_per_launch_raw": (fa + wa) * 1024.0, "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
        tot_raw += (fa + wa) * 1024.0
        tot += (2.0 * fa + wa) * 1024.0
    out["kernel"] = "big_stage_kernel + big_chain3_kernel"
    out["hbm_bytes_per_launch_raw"], out["hbm_bytes_per_launch"] = tot_raw, tot
    out["source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {tag}; per-dispatch means over the full launches of each kernel, summed over "
                     "the pair; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction, WRITE_SIZE raw")
    json.dump(out, open(os.path.join(ROOT, "profiles", "" + RND + "_cfg5_pmc.json"), "w"), indent=1)
    print(json.dumps(out))


def mfma(path, batch, N, tag):
    """fp64 matrix-core counters of the cfg-5 kernels that use them."""
    out = {"batch": batch, "N": N, "tag": tag, "kernels": {}}
    for kernel in ("big_stage_kernel", "big_chain3_kernel"):
        acc = per_dispatch(path, kernel)
        if not acc:
            continue
        full = {k: [x for x in v if x > 0.5 * max(v)] for k, v in acc.items() if v and max(v) > 0}   # full launches (not the tail passes)
        d = {k: mean(v) for k, v in full.items()}
        d["dispatches_averaged"] = {k: len(v) for k, v in full.items()}
        busy, mbusy = d.get("SQ_BUSY_CYCLES"), d.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if busy and mbusy:
            d["mfma_busy_fraction_of_sq_busy"] = mbusy / busy
        out["kernels"][kernel] = d
    out["source"] = f"rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU, {tag}; per-dispatch means over the full launches"
    json.dump(out, open(os.path.join(ROOT, "profiles", "" + RND + "_cfg5_mfma.json"), "w"), indent=1)
    print(json.dumps(out))


def sq(paths, batch, N, tag):
    out = {"batch": batch, "N": N, "tag": tag, "kernel": "lm_pass_kernel", "counters": {}}
    for pth in paths:
        acc = per_dispatch(pth, "lm_pass_kernel")
        for k, v in acc.items():
            v = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
            out["counters"][k] = mean(v)
    c = out["counters"]
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if c.get(k) is not None:
                out.setdefault("fraction_of_wave_cycles", {})[k] = c[k] / c["SQ_WAVE_CYCLES"]
    out["source"] = f"rocprofv3 --pmc (two SQ passes), {tag}; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"
    json.dump(out, open(os.path.join(ROOT, "profiles", "" + RND + "_solve_sq.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    kind = sys.argv[1]
    if kind in ("sweep", "solve"):
        traffic(kind, sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    elif kind == "cfg5":
        cfg5_traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    elif kind == "mfma":
        mfma(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    elif kind == "sq":
        sq(sys.argv[2:-3], int(sys.argv[-3]), int(sys.argv[-2]), sys.argv[-1])
    else:
        raise SystemExit(__doc__)
