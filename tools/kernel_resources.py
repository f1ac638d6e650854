"""Register / LDS / spill table of every fmx kernel from the gfx950 code object inside fmx_gpu.o (llvm-readelf --notes).
usage: python tools/kernel_resources.py [fuzzy-match_amd/csrc/fmx_gpu.o] > profiles/rNN_kernel_resources.csv"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "fuzzy-match_amd", "csrc", "fmx_gpu.o"))
tmp = tempfile.mkdtemp(prefix="fmx_co_")
try:
    shutil.copy(obj, os.path.join(tmp, "k.o"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "k.o"], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], cwd=tmp, check=True, capture_output=True, text=True).stdout
finally:
    shutil.rmtree(tmp, ignore_errors=True)
rows = []
for k in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
    def g(f):
        m = re.search(r"\." + f + r":\s*(\S+)", k)
        return m.group(1) if m else "?"
    rows.append([g("name"), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size"), g("max_flat_workgroup_size")])
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("kernel,vgpr,sgpr,vgpr_spills,sgpr_spills,scratch_bytes,static_lds_bytes,max_workgroup")
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
    if "fmx::k_" in n:
        print('"' + n.replace("fmx::", "") + '",' + ",".join(r[1:]))
