"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU against known byte counts in the access patterns of the
match() kernels (tools/microbench/pmc_calibrate.hip) and writes profiles/<round>_fetch_size_calibration.json, which
bench.py applies to the counters it collects.  Run on the GPU box:  python tools/pmc_calibrate.py [round]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "tools", "microbench", "pmc_calibrate.hip")
exe = "/tmp/pmc_calibrate"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src])


def run(counter):
    out = tempfile.mkdtemp(prefix="fmx_cal_", dir="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", exe],
                          cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            by.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    shutil.rmtree(out, ignore_errors=True)
    return {k: sorted(v)[-1][1] for k, v in by.items()}  # the second (warm) repetition


fetch = run("FETCH_SIZE")
write = run("WRITE_SIZE")
B = 8 << 30
lines = B // 128
res = {
    "_comment": "FETCH_SIZE / WRITE_SIZE are reported in KB.  factor = known HBM bytes / (counter x 1024).  Streaming kernels read "
                "(write) every byte of an 8 GiB array once.  The random kernels issue one 32 B / 16 B load into each 128 B line "
                "exactly once: `raw_bytes_per_load` is what the counter tallies per load; the random factor prices a load at a "
                "full 128 B line, the unit the memory system moves (DESIGN.md §4: random loads run at the line rate).",
    "array_bytes": B,
    "raw_kb": {"fetch": fetch, "write": write},
    "fetch_factor": {
        "stream16": B / (fetch["k_stream16"] * 1024),
        "stream8": B / (fetch["k_stream8"] * 1024),
        "random": lines * 128 / (fetch["k_random32"] * 1024),
        "random16": lines * 128 / (fetch["k_random16"] * 1024),
    },
    "raw_bytes_per_load": {"random32": fetch["k_random32"] * 1024 / lines, "random16": fetch["k_random16"] * 1024 / lines},
    # both halves of each line touched by one lane: ~ the same raw count as k_random16 = a miss fills the whole 128 B line
    # (the x2 of the streaming kernels applies to random loads too); ~ twice = 64 B sector fills (x1)
    "raw_bytes_per_line_both_halves": fetch["k_random16_pair"] * 1024 / lines,
    "write_factor": B / (write["k_write8"] * 1024),
}
path = os.path.join(ROOT, "profiles", f"{rnd}_fetch_size_calibration.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
