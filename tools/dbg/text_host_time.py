"""Design-time: host time of fmx_match_batch_text_begin / _end at C3 (does the enqueueing thread block?).
python tools/dbg/text_host_time.py [in_flight] [queries]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
if os.environ.get('IMPORT_TORCH'):
    import torch  # the wheel's bundled HIP runtime instead of /opt/rocm's
    torch.cuda.init()
import fmx_import, synth, bench
fmx = fmx_import.load().fmx
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_sent, mean_len, nq, pkw = bench.CONFIGS["C3"]
nq = int(sys.argv[2]) if len(sys.argv) > 2 else nq
L = synth.lib()
hc = L.fms_corpus_new(n_sent, float(mean_len), synth.VOCAB_TYPES, synth.CORPUS_SEED)
c_off, c_tok = synth._take(hc); vs = L.fms_vocab_size(hc)
import ctypes
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize()
wbuf = ctypes.create_string_buffer(32); words = [b"", b""]
for k in range(vs - 2):
    n = L.fms_word(k, wbuf, 32); words.append(wbuf.raw[:n])
w_offs = np.full(len(words), 0xffffffffffffffff, np.uint64); pos = 0
for k in range(2, len(words)):
    w_offs[k] = pos; pos += len(words[k]) + 1
w_arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
ix.set_forms_packed(w_arena, w_offs)
ix.to_device(0)
ix.set_option("contexts", max(2, T)); ix.set_option("max_host_active", T)
params = fmx.Params(**pkw); cap = 5
wls = []
for k in range(T + 1):
    hq = L.fms_queries_new(hc, nq, float(mean_len), synth.VOCAB_TYPES, synth.QUERY_SEED + k)
    off, tok = synth._take(hq); L.fms_free(hq)
    q_off = fmx.pinned_copy(off.astype(np.uint32)); q_tok = tok.astype(np.uint32)
    need = int(L.fms_pack_text(q_tok.ctypes.data, len(q_tok), None, 0))
    text = synth.pack_text(q_tok, out=fmx.pinned_empty(max(need, 1), np.uint8))
    wls.append((q_off, text))
slots = [(fmx.pinned_empty((nq, cap), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(T)]
STAGGER = float(os.environ.get('STAGGER_MS', '0'))
def run(k_steps, rec=None):
    def begin(i):
        a = time.perf_counter()
        h = ix.text_begin(params, wls[i % len(wls)][0], wls[i % len(wls)][1], slots[i % T], cap)
        if rec is not None: rec[0].append(time.perf_counter() - a)
        return h
    pend = {}
    for i in range(min(T, k_steps)):
        pend[i] = begin(i)
        if STAGGER: time.sleep(STAGGER / 1e3)
    for i in range(k_steps):
        a = time.perf_counter(); ix.text_end(pend.pop(i))
        if rec is not None: rec[1].append(time.perf_counter() - a); rec[2].append(time.perf_counter())
        if i + T < k_steps: pend[i + T] = begin(i + T)
run(3 * T)
if os.environ.get('NOGC'):
    import gc; gc.collect(); gc.disable()
for rep in range(int(os.environ.get('REPS', '3'))):
    time.sleep(float(os.environ.get('SLEEP_BETWEEN', '0')))
    rec = ([], [], [])
    K = int(os.environ.get('STEPS', '24'))
    t0 = time.perf_counter(); run(K, rec); dt = time.perf_counter() - t0
    b, e = np.array(rec[0]) * 1e3, np.array(rec[1]) * 1e3
    print(f"{nq} queries, {T} text calls in flight, stagger {STAGGER} ms: {dt / K * 1e3:.3f} ms/step over {K} steps; begin() median {np.median(b):.3f} max {b.max():.3f}; end() median {np.median(e):.3f} mean {e.mean():.3f} max {e.max():.3f}; steady (steps 8..24) {e[8:].mean() + b[8 - T:].mean() if len(b) > 8 else 0:.3f} ms/step")
    print("  end:  ", " ".join(f"{x:.1f}" for x in e[:24]))
    print("  done at (ms since the first begin):", " ".join(f"{(x - t0) * 1e3:.1f}" for x in rec[2]))
