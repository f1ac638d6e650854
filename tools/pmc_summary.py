"""Per-kernel sums of rocprofv3 --pmc counters for the LAST `n_last` dispatches of each kernel (the timed step)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 8
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    by[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
print("kernel,counter,dispatches_used,sum")
for k, cs in sorted(by.items()):
    if "fmx" not in k:
        continue
    for c, v in sorted(cs.items()):
        v.sort()
        last = v[-n_last:]
        print(f"{k},{c},{len(last)},{int(sum(x for _, x in last))}")
