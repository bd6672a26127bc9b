"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (full launches only: dispatches whose grid matches the largest one of that kernel).
python tools/pmc_table.py <counter_collection.csv> [name filter]"""
import csv, sys, collections, re
def short(n):
    m = re.search(r"(\w+<[^>]*>|\w+)\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[-60:]
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = collections.defaultdict(int)
for r in rows:
    k = short(r["Kernel_Name"])
    grid[k] = max(grid[k], int(r["Grid_Size"]))
for r in rows:
    k = short(r["Kernel_Name"])
    if flt and flt not in k: continue
    if int(r["Grid_Size"]) != grid[k]: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}, "dispatches", {n: len(v) for n, v in c.items()}.popitem()[1])
