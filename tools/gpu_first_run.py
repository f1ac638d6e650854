"""Bring-up script: runs a small workload stage by stage on the GPU and prints what it sees."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import fmx_import
fmx = fmx_import.load().fmx
import oracle_lib, workloads

print("devices", fmx.device_count(), flush=True)
ns, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(ns, nq, 25)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs)
t = time.time(); ix.sort(0); print("gpu sort s", time.time() - t, flush=True)
ix.finalize(); ix.to_device(0); print("on device, bytes", ix.info().device_bytes, flush=True)
o = workloads.build_oracle(c_off, c_tok, vs, sfreq=ix.array("sfreq"))
a = o.arrays()
print("sort equal:", all(np.array_equal(a[k], ix.array(k)) for k in ["suf_sid", "suf_pos", "qva"]), flush=True)
for name, kw in workloads.PARAM_SETS.items():
    cap = kw.get("nmatch", 5) or 64
    try:
        t = time.time()
        go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap, want_stats=True)
        dt = time.time() - t
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=os.cpu_count())
        bad = [q for q in range(len(oc)) if oc[q] != gc[q] or oo[q, :oc[q]].tobytes() != go[q, :gc[q]].tobytes()]
        print(name, "mismatch", len(bad), "/", len(oc), "ms", round(dt * 1e3, 2),
              {n: round(st.ms_kernel[i], 3) for i, n in enumerate(fmx.stage_names())},
              "hits", st.n_hits, "kept", st.n_kept, "cand", st.n_candidates, "dp", st.n_dp, "retries", st.n_retries, flush=True)
        for q in bad[:3]:
            print("   q", q, "oracle", oo[q, :oc[q]], "gpu", go[q, :gc[q]], flush=True)
    except Exception:
        traceback.print_exc()
