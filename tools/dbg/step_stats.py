"""Design-time: the counters of one step of a config (hits, survivors, candidates, largest survivor set, queries of the big sorter).
usage: python tools/dbg/step_stats.py [config]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fmx_import, synth, bench
fmx = fmx_import.load().fmx
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_sent, mean_len, nq, pkw = bench.CONFIGS[cfg]
L = synth.lib()
hc = L.fms_corpus_new(n_sent, float(mean_len), synth.VOCAB_TYPES, synth.CORPUS_SEED)
c_off, c_tok = synth._take(hc); vs = L.fms_vocab_size(hc)
hq = L.fms_queries_new(hc, nq, float(mean_len), synth.VOCAB_TYPES, synth.QUERY_SEED)
q_off, q_tok = synth._take(hq)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(0); ix.finalize(); ix.to_device(0)
params = fmx.Params(**pkw)
ix.match_batch(params, q_off.astype(np.uint32), q_tok.astype(np.uint32))
*_, st = ix.match_batch(params, q_off.astype(np.uint32), q_tok.astype(np.uint32), want_stats=True)
print({k: getattr(st, k) for k in ("n_hits", "n_light_hits", "n_heavy_hits", "n_heavy_pass", "n_phase_b", "n_kept", "n_candidates", "n_dp", "n_big", "max_kept", "n_retries")})
print("stage ms", {fmx.stage_name(i) if hasattr(fmx, "stage_name") else i: round(st.ms_kernel[i], 3) for i in range(13)})
