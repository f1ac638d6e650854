"""Ingestion rates on the GPU box (round 5, review item 9), one JSON object on stdout:
  text   the C2 corpus (1 M sentences, mean 20 tokens) written as a text file, then `fmx-cli -c corpus -a index` — read, tokenise,
         index (fmx_index_add_text_bulk), suffix sort on the GPU, .fmi on disk — timed by the CLI's own STEP lines
         (cli/src/FuzzyMatch-cli.cc:32-79 import loop + sort + dump)
  ids    the C3 corpus (5 M sentences) from vocabulary ids: add_ids, sort (GPU), finalize, to_device
python tools/ingest_rate.py [text_sentences] [id_sentences]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import fmx_import  # noqa: E402

fmx = fmx_import.load().fmx
n_text = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
out = {}

c_off, c_tok, _, _, vs = synth.corpus_and_queries(n_text, 10, 20)
L = synth.lib()
wbuf = C.create_string_buffer(32)
words = [b"", b""]
for k in range(vs - 2):
    m = L.fms_word(k, wbuf, 32)
    words.append(wbuf.raw[:m])
tmp = tempfile.mkdtemp(prefix="fmx_ingest_", dir=os.environ.get("TMPDIR", "/tmp"))
corpus = os.path.join(tmp, "corpus.txt")
need = int(L.fms_pack_text(c_tok.ctypes.data, len(c_tok), None, 0))
text = synth.pack_text(c_tok, out=np.empty(max(need, 1), np.uint8)).copy()
nul = np.flatnonzero(text == 0)
text[nul] = 32
text[nul[c_off[1:].astype(np.int64) - 1]] = 10
text.tofile(corpus)
cli = os.path.join(ROOT, "tools", "fmx-cli")
t0 = time.time()
r = subprocess.run([cli, "-c", corpus, "-a", "index"], capture_output=True, text=True, timeout=1200)
wall = time.time() - t0
assert r.returncode == 0, r.stderr[-2000:]
steps = []
for ln in r.stderr.splitlines():
    f = ln.split("\t")
    if len(f) >= 6 and f[0] == "STEP":
        steps.append((f[1], float(f[5])))
d = {name: t for name, t in steps}
imp = next(t for n_, t in steps if n_.startswith("Importing TM"))
srt = d["Sorting Index"]
dump = next(t for n_, t in steps if n_.startswith("Dump"))
out["text"] = {"sentences": n_text, "input_bytes": int(text.nbytes), "fmi_bytes": os.path.getsize(corpus + ".fmi"),
               "import_s": round(srt - imp, 3), "sort_s": round(dump - srt, 3), "process_wall_s": round(wall, 2),
               "read_to_fmi_on_disk_s": round(wall - imp, 2),
               "lines_per_s_import": round(n_text / max(srt - imp, 1e-9), 1),
               "note": "fmx-cli -c corpus.txt -a index (default penalty tokens tag,nbr,cas): Importing TM -> Sorting Index = read + tokenise + "
                       "index; -> Dump = suffix sort on the GPU; the rest of the process wall = .fmi written"}
for f in (corpus, corpus + ".fmi"):
    if os.path.exists(f):
        os.remove(f)
os.rmdir(tmp)

c_off, c_tok, _, _, vs = synth.corpus_and_queries(n_ids, 10, 25)
ix = fmx.Index.create()
t = {}
for name, fn in (("add_ids", lambda: ix.add_ids(c_off, c_tok, vs)), ("sort_gpu", lambda: ix.sort(0)), ("finalize", lambda: ix.finalize()),
                 ("to_device", lambda: ix.to_device(0))):
    t0 = time.time()
    fn()
    t[name] = round(time.time() - t0, 3)
out["ids"] = dict(t, sentences=n_ids, total_s=round(sum(t.values()), 3), suffixes=int(ix.info().n_suffixes))
print(json.dumps(out))
