"""Time of the suffix sort on the device at the sizes of the BASELINE configs (FMX_SORT_TIMING: HIP events around the kernels of
gpu_suffix_sort), with the hand-written sorter and with rocPRIM's merge sort under the comparator (FMX_SORTER=merge), and the
two results compared.   usage: python tools/sort_timing.py [n_sentences ...]      (default: 1000000 5000000)"""
import importlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import synth
    fmx = importlib.import_module("fuzzy-match_amd.fmx")
    n_sent = int(sys.argv[2])
    c_off, c_tok, _, _, vs = synth.corpus_and_queries(n_sent, 10, 25 if n_sent >= 5_000_000 else 20)  # the C3 / C2 corpora of bench.py
    for rep in range(2):
        ix = fmx.Index.create()
        ix.add_ids(c_off, c_tok, vs)
        t = time.time()
        ix.sort(0)
        print(f"  fmx_index_sort wall {time.time() - t:.2f}s (copies to the host included)", file=sys.stderr)
    np.save(sys.argv[3], ix.array("suf_sid"))
    np.save(sys.argv[3] + ".pos.npy", ix.array("suf_pos"))
    sys.exit(0)

out = os.environ.get("TMPDIR", "/tmp")
for n_sent in [int(a) for a in sys.argv[1:]] or [1_000_000, 5_000_000]:
    print(f"{n_sent} sentences")
    files = []
    for which in ("doubling", "merge"):
        f = os.path.join(out, f"sort_{which}.npy")
        files.append(f)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n_sent), f], capture_output=True, text=True,
                           env=dict(os.environ, FMX_SORT_TIMING="1", FMX_SORTER=which))
        print("".join("  " + l + "\n" for l in r.stderr.splitlines() if "sort" in l), end="")
        if r.returncode:
            print(r.stdout, r.stderr)
            sys.exit(1)
    same = np.array_equal(np.load(files[0]), np.load(files[1])) and np.array_equal(np.load(files[0] + ".pos.npy"), np.load(files[1] + ".pos.npy"))
    print(f"  both sorters give the same suffix array: {same}")
    if not same:
        sys.exit(1)
