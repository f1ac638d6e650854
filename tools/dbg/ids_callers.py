"""Design-time: T synchronous fmx_match_batch callers (ids in pinned host memory) on the C2 index and workload — the bench's
`pcie_inclusive_ids` leg alone, repeated; is it erratic, and does the spacing of call starts matter (FMX_HOST_SPACING_US)?
python tools/dbg/ids_callers.py [threads] [queries]"""
import os, sys, time
import concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fmx_import, synth
fmx = fmx_import.load().fmx
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(1_000_000, nq, 20)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
ix.set_option("contexts", max(2, T))
qo = fmx.pinned_copy(q_off.astype(np.uint32)); qt = fmx.pinned_copy(q_tok.astype(np.uint32))
outs = [(fmx.pinned_empty((nq, 5), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(T)]
prm = fmx.Params(fuzzy=0.7, nmatch=5)
def run(calls):
    def one(i):
        a = time.perf_counter(); ix.match_batch(prm, qo, qt, cap=5, out=outs[i % T], want_stats=bool(int(os.environ.get('WANT_STATS', '0')))); return time.perf_counter() - a
    with cf.ThreadPoolExecutor(T) as pool:
        t0 = time.perf_counter(); lat = list(pool.map(one, range(calls))); return time.perf_counter() - t0, lat
if int(os.environ.get('PRE_TEXT', '0')):   # the bench's order: text legs first
    words = [b"", b""] + [b"w%d" % k for k in range(2, vs)]
    offs = np.full(len(words), 0xffffffffffffffff, np.uint64); pos = 0
    for k in range(2, len(words)):
        offs[k] = pos; pos += len(words[k]) + 1
    ix.set_forms_packed(np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy(), offs)
    text = fmx.pinned_copy(np.frombuffer(b"".join(words[t] + b"\0" for t in q_tok), dtype=np.uint8))
    def one_t(i):
        ix.match_batch_text(prm, qo, text, cap=5, out=outs[i % T])
    with cf.ThreadPoolExecutor(T) as pool:
        t0 = time.perf_counter(); list(pool.map(one_t, range(48))); print(f"text calls first: {(time.perf_counter() - t0) / 48 * 1e3:.3f} ms/step", flush=True)
run(2 * T)
for rep in range(6):
    dt, lat = run(24)
    print(f"spacing {os.environ.get('FMX_HOST_SPACING_US', '500')} us, {T} callers, {nq} queries: {dt / 24 * 1e3:.3f} ms/step; call latency median {np.median(lat) * 1e3:.2f} max {max(lat) * 1e3:.2f} ms", flush=True)
