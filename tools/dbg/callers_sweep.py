"""Design-time: N synchronous fmx_match_batch_text callers (threads) on one index: time for a fixed number of calls.
python tools/dbg/callers_sweep.py [queries per call]"""
import os, sys, time
import concurrent.futures as cf
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fmx_import, synth
fmx = fmx_import.load().fmx
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(400_000, nq, 20)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize()
words = [b"", b""] + [b"w%d" % k for k in range(2, vs)]
offs = np.full(len(words), 0xffffffffffffffff, np.uint64); pos = 0
for k in range(2, len(words)):
    offs[k] = pos; pos += len(words[k]) + 1
ix.set_forms_packed(np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy(), offs)
ix.to_device(0)
ix.set_option("contexts", 16)
text = fmx.pinned_copy(np.frombuffer(b"".join(words[t] + b"\0" for t in q_tok), dtype=np.uint8))
qo = fmx.pinned_copy(q_off.astype(np.uint32))
outs = [(fmx.pinned_empty((nq, 5), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(16)]
prm = fmx.Params(fuzzy=0.7, nmatch=5)
calls = 64
def run(n_threads):
    def one(i):
        a = time.perf_counter()
        ix.match_batch_text(prm, qo, text, cap=5, out=outs[i % n_threads])
        return time.perf_counter() - a
    with cf.ThreadPoolExecutor(n_threads) as pool:
        list(pool.map(one, range(n_threads)))
        t0 = time.perf_counter()
        lat = list(pool.map(one, range(calls)))
        return time.perf_counter() - t0, float(np.median(lat))
for cap in (4, 16, 2):
    ix.set_option("max_host_active", cap)
    for nt in (16, 4, 8, 16, 1):
        dt, lat = run(nt)
        print(f"max_host_active {cap:2d}, {nt:2d} callers: {calls} calls of {nq} queries in {dt * 1e3:8.1f} ms = {dt / calls * 1e3:6.2f} ms/call; median latency of a call {lat * 1e3:7.2f} ms", flush=True)
