"""Design-time sweep of the host-buffer path (fmx_match_batch from pinned buffers): host_lanes x host_sub_queries,
several repeats each, on the C3 workload.  python tools/dbg/host_sweep.py [steps] [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synth  # noqa: E402
import fmx_import  # noqa: E402

fmx = fmx_import.load().fmx
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(5_000_000, 1_000_000, 25)
ix = fmx.Index.create()
ix.add_ids(c_off, c_tok, vs)
ix.sort(0)
ix.finalize()
ix.to_device(0)
params = fmx.Params(fuzzy=0.7, nmatch=5)
po, pt = fmx.pinned_copy(q_off.astype(np.uint32)), fmx.pinned_copy(q_tok.astype(np.uint32))
nq = len(q_off) - 1
out = (fmx.pinned_empty((nq, 5), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32))
def measure(tag):
    for _ in range(3):
        ix.match_batch(params, po, pt, cap=5, out=out)
    r = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ix.match_batch(params, po, pt, cap=5, out=out)
        torch.cuda.synchronize()
        r.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{tag}: ms/step " + " ".join(f"{x:.2f}" for x in r), flush=True)


for rnd in range(2):
    for hs in (262144, 524288):
        ix.set_option("host_sub_queries", hs)
        measure(f"round={rnd} hs={hs}")
