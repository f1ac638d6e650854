#!/usr/bin/env python3
"""Design-time statistics for the probe's table layout (CPU only): for every (query, start) of a synthetic batch, is the seed
5-gram in the index, how many suffixes does its group hold, how long is the seed — and from that the 128-byte lines per start
each candidate layout would touch.  usage: probe_stats.py [sentences] [queries] [mean_len]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fmx_import  # noqa: E402
import synth  # noqa: E402

fmx = fmx_import.load().fmx
n_sent = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
mean = int(sys.argv[3]) if len(sys.argv) > 3 else 25
t0 = time.time()
c_off, c_tok, q_off, q_tok, vs = synth.corpus_and_queries(n_sent, nq, mean)
ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); ix.sort(-1, 8); ix.finalize()
print(f"index: {time.time() - t0:.0f}s", flush=True)
sid, pos, buf, sp = ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos")
n = len(sid)
off = sp[sid].astype(np.int64) + pos.astype(np.int64)        # buffer index of the suffix's first token
slen = buf[sp[sid]].astype(np.int64)
rem = slen - (pos.astype(np.int64) - 1)
K = 9
tok = np.zeros((n, K), np.uint32)
for k in range(K):
    ok = rem > k
    tok[ok, k] = buf[off[ok] + k]
def h5(t):  # 64-bit hash of the first five tokens
    h = np.zeros(len(t), np.uint64)
    for k in range(5):
        h = (h ^ t[:, k].astype(np.uint64)) * np.uint64(0x9E3779B97F4A7C15)
        h ^= h >> np.uint64(29)
    return h
has5 = rem >= 5
g = h5(tok)
# groups of equal 5-grams are contiguous in the suffix array
b = np.ones(n, bool)
b[1:] = (tok[1:, :5] != tok[:-1, :5]).any(axis=1)
b &= has5
starts = np.nonzero(b)[0]
grp_hash = g[starts]
# group sizes: run until the next boundary among suffixes with >= 5 tokens that share the 5-gram
same_as_prev = np.zeros(n, bool)
same_as_prev[1:] = (tok[1:, :5] == tok[:-1, :5]).all(axis=1) & has5[1:] & has5[:-1]
run_id = np.cumsum(~same_as_prev) - 1
cnt_all = np.bincount(run_id)
grp_cnt = cnt_all[run_id[starts]]
print(f"suffixes {n}, with >= 5 tokens {int(has5.sum())}, distinct 5-grams {len(starts)} (mean group {has5.sum() / len(starts):.2f})")
for lim in (1, 2, 3, 4, 7, 8, 16, 64, 1024):
    print(f"  groups with cnt <= {lim}: {float((grp_cnt <= lim).mean()):.3f} of groups, {float(grp_cnt[grp_cnt <= lim].sum() / grp_cnt.sum()):.3f} of suffixes")
order = np.argsort(grp_hash)
gh, gc, gs = grp_hash[order], grp_cnt[order], starts[order]
# queries
p = np.diff(q_off).astype(np.int64)
msl = np.maximum(np.minimum(3, p), (np.float32(0.3) * p.astype(np.float32)).astype(np.int64))
m = np.where(p == 1, 1, np.maximum(msl, 2))
qi = np.repeat(np.arange(nq), p)
it = np.arange(len(q_tok)) - np.repeat(q_off[:-1].astype(np.int64), p)
mm, pp = m[qi], p[qi]
valid = it + mm <= pp
qt = np.zeros((len(q_tok), K), np.uint32)
for k in range(K):
    ok = it + k < pp
    idx = np.nonzero(ok)[0]
    qt[idx, k] = q_tok[idx + k]
print(f"starts {len(q_tok)}, probed {int(valid.sum())}; seed length histogram: " + " ".join(f"{k}:{float((mm[valid] == k).mean()):.3f}" for k in range(2, 13)))
v5 = valid & (mm >= 5)
qh = h5(qt)
j = np.searchsorted(gh, qh)
j[j >= len(gh)] = 0
present = v5 & (gh[j] == qh)
cnt = np.where(present, gc[j], 0)
print(f"seeds of >= 5 tokens: {float(v5.sum() / valid.sum()):.3f} of probed starts; of those the 5-gram is in the index: {float(present.sum() / v5.sum()):.3f}")
pc = cnt[present]
print("  present: group size  " + "  ".join(f"<={lim}: {float((pc <= lim).mean()):.3f}" for lim in (1, 2, 3, 4, 7, 8, 16, 64, 1024, 1 << 30)))
deep = present & (mm > 9)
print(f"  present and seed > 9 tokens: {float(deep.sum() / max(present.sum(), 1)):.3f}; present with seed == 5: {float((present & (mm == 5)).sum() / max(present.sum(), 1)):.3f}")
# lines per probed start with m >= 5 under the candidate layouts (the qva pair lives in L2 and is not counted)
lg = np.ceil(np.log2(np.maximum(pc, 1) / 8 + 1e-9)).clip(0)
need_tail = (mm[present] > 5)
v1 = 1.0 + need_tail * np.where(pc <= 8, 1 + (pc - 1) / 8.0, lg + 2)          # tails indexed by suffix, unaligned
v2 = 1.0 + need_tail * np.where(pc <= 7, 1.0, np.where(pc <= 15, 2.0, lg + 2))  # header + tails in an aligned block
absent = int(v5.sum() - present.sum())
tot = v5.sum()
print(f"lines per start (seeds >= 5): absent 1.05; v1 (bucket + unaligned tails) {(absent * 1.05 + v1.sum()) / tot:.2f}; v2 (bucket + aligned block) {(absent * 1.05 + v2.sum()) / tot:.2f}")
