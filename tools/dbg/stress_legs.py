"""Design-time: the bench's host-buffer legs (the text pipeline with calls in flight from one thread, synchronous id callers,
synchronous text callers) repeated on one index, under a watchdog — one C5 bench run of round 6 stopped answering between
"batches pinned" and the CLI leg and was killed by its timeout 23 minutes later; this is the loop that looks for it.
When an iteration makes no progress for STALL_S seconds the watchdog prints every host thread's native stack (rocgdb, if it
answers) and Python stack, and ends the process with exit code 3.
python tools/dbg/stress_legs.py [config C3|C4|C5] [iterations] [sentences] [queries]"""
import faulthandler
import os
import subprocess
import sys
import threading
import time
import concurrent.futures as cf

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fmx_import, synth

fmx = fmx_import.load().fmx
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_sent = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
STALL_S = float(os.environ.get("STALL_S", "40"))
PKW = {"C3": dict(fuzzy=0.7, nmatch=5), "C4": dict(fuzzy=0.5, nmatch=5, min_subseq_ratio=0.5, idf_penalty=1.0),
       "C5": dict(fuzzy=0.7, nmatch=5, contrast=0.3, contrast_buffer=32)}[cfg]

beat = [time.time(), "start"]


def watchdog():
    while True:
        time.sleep(2)
        if time.time() - beat[0] > STALL_S:
            sys.stderr.write(f"[stress] NO PROGRESS for {STALL_S:.0f}s in: {beat[1]}\n")
            sys.stderr.flush()
            try:
                r = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(os.getpid()), "-batch", "-ex", "thread apply all bt 14"],
                                   capture_output=True, text=True, timeout=150)
                sys.stderr.write(r.stdout[-14000:] + "\n" + r.stderr[-1500:] + "\n")
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"[stress] rocgdb: {e!r}\n")
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            sys.stderr.flush()
            os._exit(3)


L = synth.lib()
t0 = time.time()
hc = L.fms_corpus_new(n_sent, 25.0, synth.VOCAB_TYPES, synth.CORPUS_SEED)
c_off, c_tok = synth._take(hc)
vs = L.fms_vocab_size(hc)
NB = 6
raw = []
for k in range(NB):
    hq = L.fms_queries_new(hc, nq, 25.0, synth.VOCAB_TYPES, synth.QUERY_SEED + k)
    raw.append(synth._take(hq))
    L.fms_free(hq)
L.fms_free(hc)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize()
import ctypes
wbuf = ctypes.create_string_buffer(32)
words = [b"", b""]
for k in range(vs - 2):
    n = L.fms_word(k, wbuf, 32)
    words.append(wbuf.raw[:n])
w_offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
pos = 0
for k in range(2, len(words)):
    w_offs[k] = pos
    pos += len(words[k]) + 1
w_arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
ix.set_forms_packed(w_arena, w_offs)
ix.to_device(0)
ix.set_option("contexts", 6)
prm = fmx.Params(**PKW)
cap = 5


class WL:
    def __init__(self, off, tok):
        self.q_off = fmx.pinned_copy(off.astype(np.uint32)); self.q_tok = fmx.pinned_copy(tok.astype(np.uint32))
        need = int(L.fms_pack_text(self.q_tok.ctypes.data, len(self.q_tok), None, 0))
        self.text = synth.pack_text(self.q_tok, out=fmx.pinned_empty(max(need, 1), np.uint8))


wls = [WL(*r) for r in raw]
T_text, T_host = (5 if nq >= (1 << 19) else 4), 4
outs = [(fmx.pinned_empty((nq, cap), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(6)]
print(f"[stress] {cfg}: {n_sent} sentences, {NB} batches of {nq} queries ready in {time.time() - t0:.1f}s", flush=True)
threading.Thread(target=watchdog, daemon=True).start()


def tick(what):
    beat[0] = time.time(); beat[1] = what


def text_pipeline(k_steps):
    T = T_text
    ix.set_option("max_host_active", max(4, T))
    try:
        pending = {i: ix.text_begin(prm, wls[i % NB].q_off, wls[i % NB].text, outs[i % T], cap) for i in range(min(T, k_steps))}
        for i in range(k_steps):
            ix.text_end(pending.pop(i))
            tick(f"text pipeline step {i}")
            if i + T < k_steps:
                j = i + T
                pending[j] = ix.text_begin(prm, wls[j % NB].q_off, wls[j % NB].text, outs[j % T], cap)
    finally:
        ix.set_option("max_host_active", 4)


def callers(kind, calls):
    def one(i):
        wl = wls[i % NB]
        if kind == "text":
            ix.match_batch_text(prm, wl.q_off, wl.text, cap=cap, want_stats=True, out=outs[i % T_host])
        else:
            ix.match_batch(prm, wl.q_off, wl.q_tok, cap=cap, want_stats=True, out=outs[i % T_host])
        tick(f"{kind} callers, call {i}")
    with cf.ThreadPoolExecutor(T_host) as pool:
        list(pool.map(one, range(calls)))


t_all = time.time()
for it in range(iters):
    a = time.time()
    tick(f"iteration {it}: text pipeline"); text_pipeline(12)
    b_ = time.time()
    tick(f"iteration {it}: id callers"); callers("ids", 8)
    c_ = time.time()
    tick(f"iteration {it}: text callers"); callers("text", 8)
    d_ = time.time()
    if it % 10 == 0 or it == iters - 1:
        print(f"[stress] iteration {it}: pipeline {(b_ - a) / 12 * 1e3:.2f} ms/step, id callers {(c_ - b_) / 8 * 1e3:.2f}, text callers {(d_ - c_) / 8 * 1e3:.2f}", flush=True)
print(f"[stress] {iters} iterations without a stall in {time.time() - t_all:.1f}s", flush=True)
