"""Design-time: host time of fmx_match_batch_device_begin / _end for small passes (is the enqueueing thread the bottleneck?).
python tools/dbg/host_enqueue_time.py [queries] [in_flight]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import fmx_import, synth
fmx = fmx_import.load().fmx
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(5_000_000, nq, 25)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
ix.set_option("contexts", max(2, T))
dev = torch.device("cuda", 0)
d_off = torch.from_numpy(q_off.astype(np.int32)).to(dev); d_tok = torch.from_numpy(q_tok.astype(np.int32)).to(dev)
slots = [(torch.zeros((nq, 5, 4), dtype=torch.int32, device=dev), torch.zeros((nq,), dtype=torch.int32, device=dev)) for _ in range(T)]
prm = fmx.Params(fuzzy=0.7, nmatch=5)
ml = int(np.diff(q_off).max())
torch.cuda.synchronize()
def run(k_steps, rec=None):
    def begin(i):
        a = time.perf_counter()
        h = ix.device_begin(prm, nq, d_off.data_ptr(), d_tok.data_ptr(), int(q_off[-1]), slots[i % T][0].data_ptr(), slots[i % T][1].data_ptr(), 0, 5, stream=0, max_len=ml)
        if rec is not None: rec[0].append(time.perf_counter() - a)
        return h
    pend = {i: begin(i) for i in range(min(T, k_steps))}
    for i in range(k_steps):
        a = time.perf_counter()
        ix.device_end(pend.pop(i))
        if rec is not None: rec[1].append(time.perf_counter() - a)
        if i + T < k_steps: pend[i + T] = begin(i + T)
run(4 * T)
torch.cuda.synchronize()
rec = ([], [])
t0 = time.perf_counter(); run(48, rec); torch.cuda.synchronize(); dt = time.perf_counter() - t0
b, e = np.array(rec[0]) * 1e3, np.array(rec[1]) * 1e3
print(f"{nq} queries, {T} in flight: {dt / 48 * 1e3:.3f} ms/step; begin() ms median {np.median(b):.3f} mean {b.mean():.3f} max {b.max():.3f}; end() median {np.median(e):.3f} mean {e.mean():.3f} max {e.max():.3f}")
print("  begin:", " ".join(f"{x:.2f}" for x in b[:24]))
print("  end:  ", " ".join(f"{x:.2f}" for x in e[:24]))
