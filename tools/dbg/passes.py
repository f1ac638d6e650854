"""Design-time: per pass (k_setup .. next k_setup on the same stream) the stream / queue ids and the duration and start
offset of the large kernels.  python tools/dbg/passes.py <dir> [first_pass] [n]"""
import csv, glob, sys
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 0; n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(kt)))
cols = rows[0].keys()
qcol = "Queue_Id" if "Queue_Id" in cols else None
scol = "Stream_Id" if "Stream_Id" in cols else None
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", ""), r.get(qcol, ""), r.get(scol, "")) for r in rows)
setups = [i for i, e in enumerate(ev) if e[2] == "k_setup"]
big = ("k_probe", "k_expand_h", "k_expand_a", "k_expand_b<true>", "k_expand_b<false>", "k_cand_big", "k_dp_reg<32", "k_finish_pair", "k_finish<1024", "k_stats")
for pi in range(first, min(first + n, len(setups))):
    i0 = setups[pi]; t0 = ev[i0][0]; q0 = ev[i0][3]; s0 = ev[i0][4]
    line = [f"pass {pi} q{q0} s{s0}"]
    seen = set()
    for e in ev[i0:i0 + 400]:
        for b in big:
            if e[2].startswith(b) and b not in seen and e[0] >= t0 and (b != "k_probe" or e[3] == q0):
                # the first kernel of that name after this pass's k_setup on a queue this pass uses is taken as its own (approximate when passes overlap)
                seen.add(b)
                line.append(f"{b.replace('k_', '')}@{(e[0] - t0) / 1e3:.0f}+{(e[1] - e[0]) / 1e3:.0f}(q{e[3]})")
    print("  ".join(line))
