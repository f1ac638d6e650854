"""Design-time: wall time of one-at-a-time device-resident calls (with stage events) against the sum of their stage times."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fmx_import, synth
fmx = fmx_import.load().fmx
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(5_000_000, nq, 25)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
ix.set_option("contexts", 6)
dev = torch.device("cuda", 0)
d_off = torch.from_numpy(q_off.astype(np.int32)).to(dev); d_tok = torch.from_numpy(q_tok.astype(np.int32)).to(dev)
out = torch.zeros((nq, 5, 4), dtype=torch.int32, device=dev); cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
prm = fmx.Params(fuzzy=0.7, nmatch=5); ml = int(np.diff(q_off).max())
torch.cuda.synchronize()
for want_stats in (True, False, True):
    for rep in range(2):
        w = []
        for i in range(8):
            a = time.perf_counter()
            st = ix.match_batch_device(prm, nq, d_off.data_ptr(), d_tok.data_ptr(), int(q_off[-1]), out.data_ptr(), cnt.data_ptr(), 0, 5, stream=0, max_len=ml, want_stats=want_stats)
            w.append((time.perf_counter() - a) * 1e3)
        ssum = sum(st.ms_kernel) if want_stats else 0.0
        print(f"stats={want_stats}: wall per call median {np.median(w):.3f} ms (min {min(w):.3f}), stage sum of the last call {ssum:.3f} ms", flush=True)
