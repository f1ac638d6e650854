"""Shared builders/comparators for the parity tests (seeded synthetic corpora, parameter sets)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle_lib  # noqa: E402
import synth  # noqa: E402

# BASELINE.json configs (parameter sets of SURVEY §8d) + adversarial sets (SURVEY §8c "How it is used")
PARAM_SETS = {
    "C1": dict(fuzzy=0.8, nmatch=5),
    "C3": dict(fuzzy=0.7, nmatch=5),
    "C4": dict(fuzzy=0.5, nmatch=5, min_subseq_ratio=0.5, idf_penalty=1.0),
    "C5": dict(fuzzy=0.7, nmatch=5, contrast=0.3, contrast_buffer=32),
    "wide_all": dict(fuzzy=0.4, nmatch=0, min_subseq_length=2, min_subseq_ratio=0),
    "costs_101": dict(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0, costs=(1, 0, 1)),
    "costs_frac_idf": dict(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0, costs=(0.5, 1, 2),
                           idf_penalty=0.5),
    "no_perfect": dict(fuzzy=0.6, nmatch=2, no_perfect=True),
    "contrast_max": dict(fuzzy=0.5, nmatch=4, contrast=1.0, reduce=1, contrast_buffer=8, min_subseq_length=2,
                         min_subseq_ratio=0),
    "contrast_mean_all": dict(fuzzy=0.6, nmatch=0, contrast=0.5, reduce=0, contrast_buffer=4),
    "ml_negative": dict(fuzzy=0.5, nmatch=5, min_subseq_length=-1, min_subseq_ratio=0),
    "fuzzy_zero": dict(fuzzy=0.0, nmatch=10, min_subseq_length=4, min_subseq_ratio=0.2),
}


def synthetic(n_sent, n_queries, mean_len, vocab_types=synth.VOCAB_TYPES):
    return synth.corpus_and_queries(n_sent, n_queries, mean_len, vocab_types=vocab_types)


def dense_corpus(rng, n_sent, vocab, min_len=1, max_len=12):
    """Small-vocabulary corpus: lots of n-gram reuse, duplicates, one-token sentences."""
    lens = rng.integers(min_len, max_len + 1, size=n_sent)
    off = np.zeros(n_sent + 1, np.uint32)
    off[1:] = np.cumsum(lens)
    tok = rng.integers(2, 2 + vocab, size=int(off[-1])).astype(np.uint32)
    return off, tok


def edge_queries(rng, c_off, c_tok, vocab, n):
    """Copies, edits, UNK tokens (id 1), out-of-vocabulary ids, single tokens, empty and over-long patterns."""
    offs, toks = [0], []
    ns = len(c_off) - 1
    for i in range(n):
        kind = i % 9
        if kind == 0:
            s = int(rng.integers(0, ns))
            t = list(c_tok[c_off[s]:c_off[s + 1]])
        elif kind == 1:
            s = int(rng.integers(0, ns))
            t = list(c_tok[c_off[s]:c_off[s + 1]])
            for _ in range(int(rng.integers(1, 3))):
                if t:
                    t[int(rng.integers(0, len(t)))] = int(rng.integers(2, 2 + vocab))
        elif kind == 2:
            s = int(rng.integers(0, ns))
            t = list(c_tok[c_off[s]:c_off[s + 1]])
            t.insert(int(rng.integers(0, len(t) + 1)), 1)  # UNK
        elif kind == 3:
            t = [int(rng.integers(2, 2 + vocab))]
        elif kind == 4:
            t = []
        elif kind == 5:
            t = list(rng.integers(2, 2 + vocab, size=int(rng.integers(2, 20))))
        elif kind == 6:
            s = int(rng.integers(0, ns))
            a = list(c_tok[c_off[s]:c_off[s + 1]])
            s2 = int(rng.integers(0, ns))
            t = a + list(c_tok[c_off[s2]:c_off[s2 + 1]])
        elif kind == 7:
            t = [1, 1, 1]
        else:  # the empty word (id 0, also the sentence separator) inside a pattern
            s = int(rng.integers(0, ns))
            t = list(c_tok[c_off[s]:c_off[s + 1]])
            t.insert(int(rng.integers(0, len(t) + 1)), 0)
        toks.extend(int(x) for x in t)
        offs.append(len(toks))
    return np.asarray(offs, np.uint32), np.asarray(toks, np.uint32)


def build_oracle(c_off, c_tok, vocab_size, sfreq=None, nthreads=None):
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, c_tok, vocab_size, sfreq=sfreq)
    o.sort(nthreads or oracle_lib.effective_cpus())
    return o


def compare(oracle_out, oracle_cnt, got_out, got_cnt, label=""):
    """Bit-exact: same count, and per match identical (score bits, penalty bits, max_subseq, s_id) in order."""
    assert len(oracle_cnt) == len(got_cnt)
    if len(oracle_cnt) > 50_000:  # large batches: one vectorised pass first (entries beyond the count are unspecified: masked)
        oc, gc = np.asarray(oracle_cnt).astype(np.int64), np.asarray(got_cnt).astype(np.int64)
        if np.array_equal(oc, gc) and oracle_out.shape[1] == got_out.shape[1]:
            valid = np.arange(oracle_out.shape[1])[None, :] < oc[:, None]
            a = np.ascontiguousarray(oracle_out).view(np.uint32).reshape(len(oc), -1, 4)
            b = np.ascontiguousarray(got_out).view(np.uint32).reshape(len(gc), -1, 4)
            if np.array_equal(a[valid], b[valid]):
                return
    bad = []
    for q in range(len(oracle_cnt)):
        a, b = int(oracle_cnt[q]), int(got_cnt[q])
        if a != b or oracle_out[q, :a].tobytes() != got_out[q, :b].tobytes():
            bad.append(q)
    assert not bad, (f"{label}: {len(bad)} of {len(oracle_cnt)} queries differ; first: q={bad[0]} "
                     f"oracle={oracle_out[bad[0], :int(oracle_cnt[bad[0]])]} "
                     f"got={got_out[bad[0], :int(got_cnt[bad[0]])]}")
