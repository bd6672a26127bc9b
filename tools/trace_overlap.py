"""Does the result delivery of a host-driven handle overlap the next step?  Reads a rocprofv3 kernel trace (csv) and prints, for every long
copy_rows_kernel dispatch, the kernels running at the same time.   python tools/trace_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
long_copies = [r for r in rows if "copy_rows" in r[2] and r[1] - r[0] > 100_000]
for c in long_copies[-4:]:
    print("copy %.1f us  queue %s stream %s" % ((c[1] - c[0]) / 1e3, c[3], c[4]))
    for r in rows:
        if r is not c and r[0] < c[1] and r[1] > c[0]:
            print("     overlaps %-48s %.1f us (starts %+.1f us) queue %s" % (r[2], (r[1] - r[0]) / 1e3, (r[0] - c[0]) / 1e3, r[3]))
