import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import fmx_import
fmx = fmx_import.load().fmx
import oracle_lib, workloads
c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(5_000_000, 1_000_000, 25)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
kw = workloads.PARAM_SETS["C3"]
m = 60000
so, st = q_off[:m + 1], q_tok[:int(q_off[m])]
g1, c1, _, s1 = ix.match_batch(fmx.Params(**kw), so, st, cap=5, want_stats=True)
print("plane path: hits", s1.n_hits, "heavy", s1.n_heavy_hits, "pass", s1.n_heavy_pass, "retries", s1.n_retries)
ix.set_option("heavy_min", 0)
for rep in range(3):
    g3, c3, _, s3 = ix.match_batch(fmx.Params(**kw), so, st, cap=5, want_stats=True)
    bad = np.nonzero((c3 != c1) | np.any(g3.view(np.uint8).reshape(m, -1) != g1.view(np.uint8).reshape(m, -1), axis=1))[0]
    print("rep", rep, "heavy_min=0: hits", s3.n_hits, "retries", s3.n_retries, "mismatching queries", len(bad), bad[:10])
    for q in bad[:3]:
        print(q, "p", so[q+1]-so[q], "plane:", c1[q], g1[q, :c1[q]], "walk:", c3[q], g3[q, :c3[q]])
