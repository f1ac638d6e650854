"""Turns one profiling session's raw rocprofv3 output (gpurun_out/prof_<CFG>/, written by tools/profile_session.sh on the
GPU box) into the tracked summaries under profiles/: the bench line, the kernel-trace summary, and per-kernel PMC tables
(median over the warm full-size dispatches).  Usage: python tools/profile_refresh.py <CFG> [round]"""
import collections
import csv
import glob
import json
import statistics
import subprocess
import sys

cfg = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
src = f"gpurun_out/prof_{cfg}"
low = cfg.lower()


def load(pat):
    f = sorted(glob.glob(pat, recursive=True))
    if not f:
        return None
    by = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f[-1])):
        by[r["Kernel_Name"].split("(")[0]][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return by


line = open(f"{src}/bench.json").read().strip().split("\n")[-1]
json.loads(line)
open(f"profiles/{rnd}_{low}_bench_1gpu.json", "w").write(line + "\n")

kt = sorted(glob.glob(f"{src}/kt/**/*kernel_trace.csv", recursive=True))[-1]
open(f"profiles/{rnd}_{low}_kernel_trace_summary.csv", "w").write(
    subprocess.run([sys.executable, "tools/kernel_trace_summary.py", kt], capture_output=True, text=True).stdout)

import os
import shutil
if os.path.exists(f"{src}/pipeline_busy.txt"):
    shutil.copy(f"{src}/pipeline_busy.txt", f"profiles/{rnd}_{low}_pipeline_busy.txt")

names = ["SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
         "SQ_INSTS_LDS"]
by = load(f"{src}/pmc_SQ/**/*counter_collection.csv")
if by:
    with open(f"profiles/{rnd}_{low}_pmc_sq.csv", "w") as fo:
        fo.write("kernel,dispatches_used,SQ_BUSY_CYCLES_per_SE," + ",".join(names[1:]) + ",valu_busy_frac,wait_frac,waves_per_simd,valu_per_wave\n")
        for k in sorted(by):
            if "fmx" not in k or "k_dv_" in k or "k_build" in k or "sort_keys" in k or "extract_keys" in k:
                continue
            mx = max(c.get("SQ_WAVES", 0) for c in by[k].values())
            full = [c for _, c in sorted(by[k].items()) if c.get("SQ_WAVES", 0) >= 0.98 * mx]
            if len(full) > 2:
                full = full[2:]  # the first pass runs while the workspace grows (cold)
            g = {n: statistics.median(c.get(n, 0.0) for c in full) for n in names}
            busy = g["SQ_BUSY_CYCLES"] / 32
            if busy <= 0:
                continue
            fo.write(f"\"{k}\",{len(full)},{int(busy)}," + ",".join(str(int(g[n])) for n in names[1:]) +
                     f",{g['SQ_INSTS_VALU'] * 4 / 1024 / busy:.3f},{g['SQ_WAIT_ANY'] / max(g['SQ_WAVE_CYCLES'], 1):.3f},"
                     f"{g['SQ_WAVE_CYCLES'] * 4 / 1024 / busy:.2f},{g['SQ_INSTS_VALU'] / max(g['SQ_WAVES'], 1):.0f}\n")
for cname in ["FETCH_SIZE", "WRITE_SIZE"]:
    byc = load(f"{src}/pmc_{cname}/**/*counter_collection.csv")
    if not byc:
        continue
    with open(f"profiles/{rnd}_{low}_pmc_{cname.lower()}.csv", "w") as fo:
        fo.write("kernel,counter,dispatches_used,median_value_kb\n")
        for k in sorted(byc):
            if "fmx" not in k or "k_dv_" in k or "k_build" in k or "sort_keys" in k or "extract_keys" in k:
                continue
            vals = [c[cname] for _, c in sorted(byc[k].items())]
            top = sorted(vals)
            ref = top[-3] if len(top) > 3 else top[-1]
            full = [v for v in vals if 0.8 * ref <= v <= 1.25 * ref]
            fo.write(f"\"{k}\",{cname},{len(full)},{int(statistics.median(full))}\n")
print("profiles refreshed for", cfg)
