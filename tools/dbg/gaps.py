"""Design-time: idle gaps of the GPU (> min_us between the end of everything running and the next kernel start) in a
rocprofv3 kernel trace, with the k_probe index they fall behind.  python tools/dbg/gaps.py <dir> [min_us]"""
import csv, glob, sys
d = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", "")) for r in csv.DictReader(open(kt)))
t0 = rows[0][0]; ce = None; last = None; n_probe = 0
for s, e, n in rows:
    if n.startswith("k_probe"):
        n_probe += 1
        print(f"probe {n_probe:3d} at {(s - t0) / 1e6:10.3f} ms")
    if ce is not None and s > ce + min_us * 1e3:
        print(f"   gap {(s - ce) / 1e3:9.1f} us at {(ce - t0) / 1e6:10.3f} ms after {last} before {n}")
    if ce is None or e > ce: ce, last = e, n
