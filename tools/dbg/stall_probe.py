"""Design-time: the value leg (text batches from one host thread, T in flight) repeated in one process: ms/step of every repetition.
usage: stall_probe.py [T] [reps] [steps]   (environment: GPU_MAX_HW_QUEUES, FMX_STREAM_CLASSES, FMX_HOST_INLINE_COPY ...)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fmx_import, synth
fmx = fmx_import.load().fmx
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = synth.lib()
hc = L.fms_corpus_new(5_000_000, 25.0, synth.VOCAB_TYPES, synth.CORPUS_SEED)
c_off, c_tok = synth._take(hc)
vs = L.fms_vocab_size(hc)
batches = []
for k in range(T + 1):
    hq = L.fms_queries_new(hc, 1_000_000, 25.0, synth.VOCAB_TYPES, synth.QUERY_SEED + k)
    off, tok = synth._take(hq); L.fms_free(hq)
    text = synth.pack_text(tok, out=fmx.pinned_empty(int(L.fms_pack_text(tok.ctypes.data, len(tok), None, 0)), np.uint8))
    batches.append((fmx.pinned_copy(off), text))
L.fms_free(hc)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize()
import ctypes
buf = ctypes.create_string_buffer(32); words = [b"", b""]
for k in range(vs - 2):
    n = L.fms_word(k, buf, 32); words.append(buf.raw[:n])
offs = np.full(len(words), 0xffffffffffffffff, np.uint64); pos = 0
for k in range(2, len(words)):
    offs[k] = pos; pos += len(words[k]) + 1
arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
ix.set_forms_packed(arena, offs)
ix.to_device(0); ix.set_option("contexts", max(T, 2))
prm = fmx.Params(fuzzy=0.7, nmatch=5)
nq = 1_000_000
outs = [(fmx.pinned_empty((nq, 5), fmx.MATCH_DTYPE), fmx.pinned_empty(nq, np.uint32), fmx.pinned_empty(nq, np.uint32)) for _ in range(T)]
def run(k_steps):
    pend = {i: ix.text_begin(prm, batches[i % (T + 1)][0], batches[i % (T + 1)][1], outs[i % T], 5) for i in range(min(T, k_steps))}
    for i in range(k_steps):
        ix.text_end(pend.pop(i))
        if i + T < k_steps:
            j = i + T
            pend[j] = ix.text_begin(prm, batches[j % (T + 1)][0], batches[j % (T + 1)][1], outs[j % T], 5)
run(2 * T)
res = []
for r in range(reps):
    t = time.perf_counter(); run(steps); res.append((time.perf_counter() - t) / steps * 1e3)
print(f"T={T} env={ {k: os.environ.get(k) for k in ('GPU_MAX_HW_QUEUES','FMX_STREAM_CLASSES','FMX_HOST_INLINE_COPY','FMX_SHARED_COPY')} } ms/step per repetition: " + " ".join(f"{x:.2f}" for x in res), flush=True)
