"""Builds the C2 index, saves it as .fmi, writes the queries as id lines, runs tools/one_rate for several thread counts,
and times the oracle one pattern at a time on the host cores for comparison."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth, fmx_import, oracle_lib
fmx = fmx_import.load().fmx
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(1_000_000, 100_000, 20)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize()
ix.save("/tmp/c2.fmi")
with open("/tmp/c2_q.txt", "w") as f:
    for q in range(20000):
        f.write(" ".join(str(int(x)) for x in q_tok[q_off[q]:q_off[q + 1]]) + "\n")
o = oracle_lib.OracleIndex(); o.add_ids_bulk(c_off, c_tok, vs, sfreq=ix.array("sfreq")); o.sort(oracle_lib.effective_cpus())
n = 20000; t0 = time.perf_counter()
o.match_batch(oracle_lib.Params(fuzzy=0.7, nmatch=5), q_off[:n + 1], q_tok[:int(q_off[n])], 5, nthreads=oracle_lib.effective_cpus())
print(f"oracle, {oracle_lib.effective_cpus()} threads: {n / (time.perf_counter() - t0):.0f} patterns/s", flush=True)
for nt in sys.argv[1:]:
    subprocess.run([os.path.join(ROOT, "tools", "one_rate"), "/tmp/c2.fmi", "/tmp/c2_q.txt", nt, "3"])
