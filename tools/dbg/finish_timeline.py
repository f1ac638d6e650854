"""Timeline of the finish stage of the LAST pass in a rocprofv3 kernel trace (csv): start / end in us relative to k_finish_pair."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", "") for r in rows]
last = max(i for i, n in enumerate(names) if n.startswith("k_finish_pair"))
t0 = int(rows[last]["Start_Timestamp"])
for r, n in zip(rows[last - 2:], names[last - 2:]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} {n[:60]}")
    if n.startswith("k_stats"):
        break
