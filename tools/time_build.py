"""Design-time: wall time of each index-build step on the C3 corpus (5 M sentences).  python tools/time_build.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
import fmx_import  # noqa: E402

fmx = fmx_import.load().fmx
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(5_000_000, 10, 25)


def step(name, fn):
    t0 = time.time()
    r = fn()
    print(f"{name} {time.time() - t0:.2f}s", flush=True)
    return r


ix = fmx.Index.create()
step("add_ids", lambda: ix.add_ids(c_off, c_tok, vs))
step("sort (GPU)", lambda: ix.sort(0))
step("finalize", lambda: ix.finalize())
step("to_device (derive on the GPU)", lambda: ix.to_device(0))
path = "/tmp/c3.fmi"
step("save", lambda: ix.save(path))
print("fmi bytes", os.path.getsize(path))
ix2 = step("load", lambda: fmx.Index.load(path))
step("finalize", lambda: ix2.finalize())
step("to_device (derive on the GPU)", lambda: ix2.to_device(0))
os.remove(path)
