"""Design-time: the START of the value leg's timed region (calls begun on an idle GPU): every copy and every kernel of at least
min_us from the last long idle gap of the trace on.  python tools/dbg/fill_timeline.py <dir> [min_us] [ms to show]"""
import csv, glob, sys
d = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0; show_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", ""), r.get("Stream_Id", "")))
mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
if mc:
    for r in csv.DictReader(open(mc[-1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", ""), ""))
rows.sort()
# the last idle gap of more than 2 ms: the synchronisation in front of the timed region
end = 0; start_i = 0
for i, r in enumerate(rows):
    if end and r[0] - end > 2_000_000: start_i = i
    end = max(end, r[1])
t0 = rows[start_i][0]
for s, e, n, st in rows[start_i:]:
    if (s - t0) / 1e6 > show_ms: break
    if (e - s) / 1e3 >= min_us or n.startswith("k_setup") or n.startswith("COPY"):
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n[:48]:48s} {st}")
