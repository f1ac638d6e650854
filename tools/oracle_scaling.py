"""Thread scaling of the CPU oracle on this host (diagnostic for bench.py's cpu_baseline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_lib, synth
ns, nq, ml = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(ns, nq, ml)
o = oracle_lib.OracleIndex(); o.add_ids_bulk(c_off, c_tok, vs)
t = time.time(); o.sort(os.cpu_count()); print("sort", round(time.time() - t, 1), "s", flush=True)
prm = oracle_lib.Params(fuzzy=0.7, nmatch=5)
for nt in [1, 8, 32, 64, 128, 256]:
    if nt > (os.cpu_count() or 1): break
    n = min(nq, 4000 * nt)
    t = time.time(); o.match_batch(prm, q_off[:n + 1], q_tok[:int(q_off[n])], 5, nthreads=nt); dt = time.time() - t
    print(nt, "threads", round(n / dt), "q/s", round(n / dt / nt), "per thread", flush=True)
