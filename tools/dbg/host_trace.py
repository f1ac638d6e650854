"""Design-time: a few fmx_match_batch calls (pinned host buffers) on the C3 workload, to be run under
rocprofv3 --kernel-trace --memory-copy-trace.  python tools/dbg/host_trace.py <lanes> <host_sub_queries> [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
import fmx_import  # noqa: E402

fmx = fmx_import.load().fmx
lanes, hs = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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
ix.set_option("host_lanes", lanes)
ix.set_option("host_sub_queries", hs)
for _ in range(3 + steps):
    ix.match_batch(params, po, pt, cap=5, out=out)
