"""Design-time: GPU busy time of the host-buffer path with T calls in flight.  Run under
rocprofv3 --kernel-trace --memory-copy-trace; then: python tools/host_busy.py analyse <dir>"""
import csv, glob, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if sys.argv[1] == "analyse":
    d = sys.argv[2]
    kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmx::", "")) for r in csv.DictReader(open(kt))]
    rows.sort()
    probes = [r for r in rows if r[2].startswith("k_probe")]
    n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    t0 = probes[-n_last][0]; t1 = max(r[1] for r in rows)
    sel = [r for r in rows if r[0] >= t0]
    # union busy
    busy = 0; cur_s, cur_e = None, None
    for s, e, _ in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t1 - t0
    mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
    copies = []
    if mc:
        for r in csv.DictReader(open(mc[-1])):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[12:] ))
    ev = sorted(sel + [c for c in copies if c[0] >= t0])
    # idle gaps of the kernels (> 40 us) with what ended before / started after, and the copies in flight then
    ce = None; last = None
    for s_, e_, n_ in sel:
        if ce is not None and s_ > ce + 40000:
            inflight = [c[2] for c in copies if c[0] < s_ and c[1] > ce]
            print(f"  gap {(s_-ce)/1e3:7.1f} us at {(ce-t0)/1e6:8.3f} ms after {last:28s} before {n_:24s} copies in flight: {inflight}")
        if ce is None or e_ > ce: ce, last = e_, n_
    print(f"window {wall/1e6:.2f} ms for {n_last} passes = {wall/1e6/n_last:.3f} ms/pass; GPU busy (union of kernels) {busy/1e6:.2f} ms = {100*busy/wall:.1f}%")
    if len(sys.argv) > 4:  # timeline of the first ~N ms of the window
        lim = float(sys.argv[4]) * 1e6
        for s_, e_, n_ in ev:
            if s_ - t0 > lim: break
            if e_ - s_ > 60000: print(f"    {(s_-t0)/1e6:8.3f} {(e_-t0)/1e6:8.3f} {(e_-s_)/1e3:8.1f}  {n_}")
    agg = {}
    for s, e, n in sel: agg[n] = agg.get(n, 0) + (e - s)
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]: print(f"  {n:40s} {v/1e6/n_last:8.3f} ms/pass (sum of durations)")
    sys.exit(0)
import synth, fmx_import
fmx = fmx_import.load().fmx
T = int(sys.argv[1]); steps = int(sys.argv[2])
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(5_000_000, int(os.environ.get('NQ', 1_000_000)), 25)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
ix.set_option("contexts", max(2, T))
for opt in ("FMX_TURNS", "FMX_HOST_SUB_QUERIES"):
    if os.environ.get(opt): ix.set_option(opt[4:].lower(), int(os.environ[opt]))
params = fmx.Params(fuzzy=0.7, nmatch=5)
po, pt = fmx.pinned_copy(q_off.astype(np.uint32)), fmx.pinned_copy(q_tok.astype(np.uint32))
nq = len(q_off) - 1
outs = [(fmx.pinned_empty((nq, 5), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(T)]
def worker(k):
    for _ in range(steps): ix.match_batch(params, po, pt, cap=5, out=outs[k])
for rnd in range(2):
    th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    print(f"round {rnd}: {(time.perf_counter()-t0)/(T*steps)*1e3:.3f} ms/step", flush=True)
