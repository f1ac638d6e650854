"""Turns one profiling session's raw rocprofv3 output (gpurun_out/<tag>*) into the tracked summaries under
profiles/: bench lines, kernel-trace summary, PMC tables (median over the warm full-size dispatches), and
the k_probe traffic file bench.py reads.  Usage: python tools/profile_refresh.py <tag> [round]"""
import collections
import csv
import glob
import json
import statistics
import subprocess
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"


def load(pat):
    f = sorted(glob.glob(pat))[-1]
    by = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        by[r["Kernel_Name"].split("(")[0]][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return by


for c in ["C3", "C2", "C4", "C5"]:
    try:
        line = open(f"gpurun_out/bench_{c}.json").read().strip().split("\n")[-1]
        json.loads(line)
        open(f"profiles/{rnd}_{c.lower()}_bench_1gpu.json", "w").write(line + "\n")
    except Exception as e:  # noqa: BLE001
        print("skip", c, e)

kt = sorted(glob.glob(f"gpurun_out/kt{tag}/*/*kernel_trace.csv"))[-1]
open(f"profiles/{rnd}_c3_kernel_trace_summary.csv", "w").write(
    subprocess.run([sys.executable, "tools/kernel_trace_summary.py", kt], capture_output=True, text=True).stdout)

names = ["SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
         "SQ_INSTS_LDS"]
by = load(f"gpurun_out/pmc{tag}_SQ/*/*counter_collection.csv")
with open(f"profiles/{rnd}_c3_pmc_sq.csv", "w") as fo:
    fo.write("kernel,dispatches_used,SQ_BUSY_CYCLES_per_SE," + ",".join(names[1:]) + ",valu_busy_frac,wait_frac,waves_per_simd\n")
    for k in sorted(by):
        if "fmx" not in k:
            continue
        mx = max(c.get("SQ_WAVES", 0) for c in by[k].values())
        full = [c for _, c in sorted(by[k].items()) if c.get("SQ_WAVES", 0) >= 0.98 * mx]
        if len(full) > 2:
            full = full[2:]  # the first pass runs twice while the workspace grows (cold)
        g = {n: statistics.median(c.get(n, 0.0) for c in full) for n in names}
        busy = g["SQ_BUSY_CYCLES"] / 32
        if busy <= 0:
            continue
        fo.write(f"\"{k}\",{len(full)},{int(busy)}," + ",".join(str(int(g[n])) for n in names[1:]) +
                 f",{g['SQ_INSTS_VALU'] * 4 / 1024 / busy:.3f},{g['SQ_WAIT_ANY'] / max(g['SQ_WAVE_CYCLES'], 1):.3f},"
                 f"{g['SQ_WAVE_CYCLES'] * 4 / 1024 / busy:.2f}\n")
out = {}
for cname in ["FETCH_SIZE", "WRITE_SIZE"]:
    byc = load(f"gpurun_out/pmc{tag}_{cname}/*/*counter_collection.csv")
    with open(f"profiles/{rnd}_c3_pmc_{cname.lower()}.csv", "w") as fo:
        fo.write("kernel,counter,dispatches_used,median_value_kb\n")
        for k in sorted(byc):
            if "fmx" not in k:
                continue
            vals = [c[cname] for _, c in sorted(byc[k].items())]
            top = sorted(vals)
            ref = top[-3] if len(top) > 3 else top[-1]
            full = [v for v in vals if 0.8 * ref <= v <= 1.25 * ref]
            v = statistics.median(full)
            fo.write(f"\"{k}\",{cname},{len(full)},{int(v)}\n")
            out[(k, cname)] = v
raw, w = out[("fmx::k_probe", "FETCH_SIZE")], out[("fmx::k_probe", "WRITE_SIZE")]
tp = f"profiles/{rnd}_c3_probe_traffic.json"
d = json.load(open(tp))
d.update({"fetch_kb_raw_per_step": int(raw), "write_kb_per_step": int(w), "fetch_bytes_per_launch_raw": int(raw) * 1024,
          "fetch_bytes_per_launch_corrected": int(raw) * 2048, "write_bytes_per_launch": int(w) * 1024,
          "traffic_bytes_per_launch": int(raw) * 2048 + int(w) * 1024})
json.dump(d, open(tp, "w"), indent=1)
print("k_probe traffic per launch: %.2f GB" % (d["traffic_bytes_per_launch"] / 1e9))
for k in ["fmx::k_expand_a", "fmx::k_expand_b", "fmx::k_probe"]:
    print(k, "fetch GB (x2) %.2f" % (out[(k, "FETCH_SIZE")] * 2048 / 1e9), "write GB %.2f" % (out[(k, "WRITE_SIZE")] * 1024 / 1e9))
