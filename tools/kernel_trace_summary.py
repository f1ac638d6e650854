"""Summarise a rocprofv3 --kernel-trace CSV into a per-kernel table (calls, total/avg/min/max ns, share)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in agg.values())
import statistics
print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent,median_ns")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k},{len(v)},{sum(v)},{sum(v) // len(v)},{min(v)},{max(v)},{100.0 * sum(v) / total:.2f},{int(statistics.median(v))}")
