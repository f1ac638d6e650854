"""Design-time: four host threads x device calls on their own streams, many rounds; reports every call whose records differ."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("FMX_SMALL_PATH", "0")
import numpy as np
import fmx_import, oracle_lib, workloads
from test_gpu_device_entry import Hip, _records
fmx = fmx_import.load().fmx
hip = Hip()
c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(200_000, 60_000, 20)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0); ix.set_option("contexts", 4)
kw = dict(fuzzy=0.7, nmatch=5)
ref = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=5)
nq, cap = len(q_off) - 1, 5
d_off, d_tok = hip.upload(q_off.astype(np.uint32)), hip.upload(q_tok.astype(np.uint32))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = []
def worker(k, rnd):
    st = hip.stream()
    for call in range(3):
        d_out, d_cnt = hip.zeros(nq * cap * 16), hip.zeros(nq * 4)
        ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), d_out, d_cnt, 0, cap, stream=st, max_len=int(np.diff(q_off).max()), want_stats=False)
        go, gc = _records(hip, d_out, d_cnt, nq, cap)
        diff = int((gc != ref[1]).sum())
        if diff or any(go[q, :gc[q]].tobytes() != ref[0][q, :ref[1][q]].tobytes() for q in range(0, nq, 97)):
            bad.append((rnd, k, call, diff, int(gc.sum()), int(ref[1].sum()), int((gc[:30000] != ref[1][:30000]).sum())))
for rnd in range(rounds):
    th = [threading.Thread(target=worker, args=(k, rnd)) for k in range(4)]
    for t in th: t.start()
    for t in th: t.join()
print("bad calls (round, thread, call, counts differing, sum got, sum ref, differing in first half):", bad, flush=True)
