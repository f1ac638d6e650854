"""Design-time: histograms of the light queries of one C3/C4/C5 step (hits, survivors of dominance + length filter, kept after
coverage, by pattern class).  Needs a library built with -DFMX_LIGHT_STATS (tools/build_variant.sh lstats -DFMX_LIGHT_STATS=1).
usage: FMX_LIB=$PWD/build/libfmx_lstats.so python tools/dbg/light_stats.py [config] > gpurun_out/light_stats.json"""
import ctypes, json, os, sys
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
lib = ctypes.CDLL(fmx.LIB_PATH)
buf = (ctypes.c_ulonglong * 1024)()
ix.match_batch(params, q_off.astype(np.uint32), q_tok.astype(np.uint32))   # warm-up: sizes learned, passes may be re-run
assert lib.fmx_debug_light_stats(None, 1) == 0
ix.match_batch(params, q_off.astype(np.uint32), q_tok.astype(np.uint32))
assert lib.fmx_debug_light_stats(buf, 0) == 0
h = np.array(buf[:], dtype=np.int64)
out = {"config": cfg, "light_queries": int(h[0:257].sum()),
       "hits": h[0:257].tolist(), "survivors": h[300:365].tolist(), "kept": h[400:465].tolist(),
       "pclass_x_survivors": h[500:532].reshape(4, 8).tolist(), "survivors_x_kept": h[540:604].reshape(8, 8).tolist(),
       "p": h[610:711].tolist(), "pclass_x_kept": h[720:752].reshape(4, 8).tolist(),
       "buckets": ["0", "1", "2", "3", "4", "5-8", "9-16", ">16"], "pclasses": ["<=16", "<=32", "<=64", ">64"]}
print(json.dumps(out))
