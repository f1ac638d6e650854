"""End-to-end throughput of tools/fmx-cli on a synthetic tokenised corpus (SURVEY §8d generator spelled out as
text): builds the .fmi with `-a index`, then times `-a match` on the query file (stdin -> stdout).
Usage: python tools/cli_throughput.py [n_sentences] [n_queries] [batch,batch,...]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402

n_sent = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
out_dir = os.environ.get("TMPDIR", "/tmp")
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(n_sent, n_q, 20)
words = np.array([synth.word(k) for k in range(vs)], dtype=object)


def dump(path, off, tok):
    w = words[tok]
    with open(path, "w") as f:
        f.write("\n".join(" ".join(w[off[i]:off[i + 1]]) for i in range(len(off) - 1)))
        f.write("\n")


corpus, queries = os.path.join(out_dir, "cli_corpus.tok"), os.path.join(out_dir, "cli_queries.tok")
dump(corpus, c_off, c_tok)
dump(queries, q_off, q_tok)
cli = os.path.join(ROOT, "tools", "fmx-cli")
t = time.time()
subprocess.check_call([cli, "--tokenised", "-c", corpus, "-a", "index"], stderr=subprocess.DEVNULL)
print(f"index: {n_sent} sentences in {time.time() - t:.1f}s")
for batch in ([int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else (32768, 65536, 131072)):
    t = time.time()
    r = subprocess.run([cli, "--tokenised", "-i", corpus + ".fmi", "-a", "match", "-f", "0.7", "-n", "5", "--batch", str(batch)],
                       stdin=open(queries), stdout=open(os.path.join(out_dir, "cli_out.tsv"), "w"), stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, FMX_CLI_TIMING="1"))
    dt = time.time() - t
    steps = {l.split("\t")[1]: float(l.split("\t")[5]) for l in r.stderr.splitlines() if l.startswith("STEP")}
    match_s = steps.get("Done", dt) - steps.get("Matching", 0.0)
    print(f"match: {n_q} patterns, batch {batch}: wall {dt:.2f}s (index load included), matching phase {match_s:.3f}s = "
          f"{n_q / max(match_s, 1e-9) / 1e3:.0f} k patterns/s; " + " | ".join(l for l in r.stderr.splitlines() if l.startswith(("NMATCH", "TIMING"))))
