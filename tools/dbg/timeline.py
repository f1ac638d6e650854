"""Design-time: GPU timeline of the last calls in a rocprofv3 kernel + memory-copy trace: per kernel launch
start/end relative to the first, grouped into passes (k_setup starts a pass).  python tools/dbg/timeline.py <dir> [n_passes] [min_us]"""
import csv
import glob
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", ""), "K", r.get("Stream_Id", "")))
mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
if mc:
    for r in csv.DictReader(open(mc[-1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", "")), "C", ""))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_setup"]
first = starts[-n_last] if len(starts) >= n_last else 0
t0 = rows[first][0]
for s, e, n, kind, st in rows[first:]:
    dur = (e - s) / 1e3
    if dur < min_us and kind == "K" and n not in ("k_setup",):
        continue
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {dur:9.1f}  {n[:60]} {st}")
