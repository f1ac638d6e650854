"""ctypes wrapper of oracle/libfm_oracle.so — the CPU restatement of the reference's match() path.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only.  Nothing under fuzzy-match_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libfm_oracle.so")

MATCH_DTYPE = np.dtype([("score", np.float32), ("penalty", np.float32),
                        ("max_subseq", np.int32), ("s_id", np.uint32)])

COUNTER_FIELDS = ["equal_range_calls", "compare_steps", "probe_bytes", "suffix_visits", "suffix_kept",
                  "reg_bytes", "candidates", "cand_bytes", "coverage_tokens", "dp_calls", "dp_cells",
                  "out_bytes", "queries", "queries_with_match"]


class Params(C.Structure):
    _fields_ = [("fuzzy", C.c_float), ("nmatch", C.c_uint32), ("no_perfect", C.c_int32),
                ("min_subseq_length", C.c_int32), ("min_subseq_ratio", C.c_float),
                ("idf_penalty", C.c_float), ("insert_cost", C.c_float), ("delete_cost", C.c_float),
                ("replace_cost", C.c_float), ("contrast", C.c_float), ("reduce", C.c_int32),
                ("contrast_buffer", C.c_int32)]

    def __init__(self, fuzzy=0.8, nmatch=5, no_perfect=False, min_subseq_length=3,
                 min_subseq_ratio=0.3, idf_penalty=0.0, costs=(1.0, 1.0, 1.0), contrast=0.0,
                 reduce=0, contrast_buffer=-1):
        super().__init__(fuzzy, nmatch, int(no_perfect), min_subseq_length, min_subseq_ratio,
                         idf_penalty, costs[0], costs[1], costs[2], contrast, reduce, contrast_buffer)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


_lib = None


def effective_cpus():
    """Hardware threads this process may really use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.fmo_index_create.restype = vp
    L.fmo_index_create.argtypes = [u64]
    L.fmo_index_free.argtypes = [vp]
    L.fmo_index_add.argtypes = [vp, vp, u32, C.c_char_p, u32, C.POINTER(u64), C.POINTER(C.c_char_p)]
    L.fmo_index_add_bulk.argtypes = [vp, vp, vp, u32]
    L.fmo_index_set_vocab.argtypes = [vp, u32, vp]
    L.fmo_index_sort.argtypes = [vp, C.c_int]
    L.fmo_index_set_arrays.argtypes = [vp, vp, vp, u64, vp, u64, vp, u64, vp, u64]
    L.fmo_index_set_real.argtypes = [vp, u32, C.c_char_p, u32, C.POINTER(u64), C.POINTER(C.c_char_p)]
    for name in ["num_suffixes", "num_sentences", "buffer_size", "qva_size"]:
        f = getattr(L, "fmo_index_" + name)
        f.restype = u64
        f.argtypes = [vp]
    for name in ["suf_sid", "suf_pos", "buffer", "sentence_pos", "qva"]:
        f = getattr(L, "fmo_index_" + name)
        f.restype = vp
        f.argtypes = [vp]
    L.fmo_equal_range.argtypes = [vp, vp, u64, u64, u64, C.POINTER(u64 * 2), C.POINTER(Counters)]
    L.fmo_match.restype = C.c_int64
    L.fmo_match.argtypes = [vp, C.POINTER(Params), vp, u32, C.c_char_p, u32, C.POINTER(u64),
                            C.POINTER(C.c_char_p), vp, u64, C.POINTER(Counters)]
    L.fmo_match_batch.argtypes = [vp, C.POINTER(Params), vp, vp, u64, C.c_int, vp, vp, u64, C.POINTER(Counters)]
    L.fmo_match_batch_real.argtypes = [vp, C.POINTER(Params), vp, vp, u64, vp, vp, C.c_int, vp, vp, u64, C.POINTER(Counters)]
    L.fmo_index_set_real_bulk.argtypes = [vp, u32, u32, vp, vp]
    L.fmo_index_count_vocab.argtypes = [vp, u32]
    L.fmo_subsequence.argtypes = [vp, vp, u32, C.c_char_p, u32, C.POINTER(u64), C.POINTER(C.c_char_p), u32, C.c_int, C.c_int,
                                  C.c_float, C.c_int, vp, C.POINTER(u32)]
    L.fmo_edit_distance_ids.restype = C.c_float
    L.fmo_edit_distance_ids.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.fmo_edit_distance_char.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.fmo_normalizer.restype = C.c_float
    L.fmo_normalizer.argtypes = [u64, u64, C.c_float, C.c_float, C.c_float]
    L.fmo_score_from_cost.restype = C.c_float
    L.fmo_score_from_cost.argtypes = [C.c_float]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _itok_args(itoks):
    itoks = itoks or {}
    n = len(itoks)
    idx = (C.c_uint64 * max(n, 1))(*list(itoks.keys()))
    strs = (C.c_char_p * max(n, 1))(*[v.encode("utf-8") for v in itoks.values()])
    return n, idx, strs


def tab_joined(stream, off):
    """A NUL-separated spelling stream (one spelling per token, CSR offsets `off` in tokens) as the per-sentence tab-joined,
    NUL-terminated strings the oracle takes (fuzzy::Sentence's tokstring).  Returns (packed uint8 array, uint64 byte offsets)."""
    stream = np.array(stream, dtype=np.uint8, copy=True)
    off = np.asarray(off, dtype=np.int64)
    nul = np.flatnonzero(stream == 0)
    assert len(nul) == off[-1], (len(nul), off[-1])
    stream[nul] = 9
    nonempty = off[1:] > off[:-1]
    stream[nul[off[1:][nonempty] - 1]] = 0
    packed = np.concatenate([stream, np.zeros(1, np.uint8)])
    starts = np.concatenate([np.zeros(1, np.int64), nul + 1])   # byte where token t starts
    boff = np.where(nonempty, starts[np.minimum(off[:-1], len(starts) - 1)], len(packed) - 1).astype(np.uint64)
    return packed, boff


class Vocab:
    """VocabIndexer (src/vocab_indexer.cc): id 0 = "", id 1 = ｟unk｠, words from 2."""

    def __init__(self):
        self.forms = ["", "｟unk｠"]
        self.sfreq = [0, 0]
        self.index = {"": 0, "｟unk｠": 1}

    def add_words(self, tokens):
        ids = []
        for t in tokens:
            if t not in self.index:
                self.index[t] = len(self.forms)
                self.forms.append(t)
                self.sfreq.append(0)
            ids.append(self.index[t])
        for i in set(ids):
            self.sfreq[i] += 1
        return ids

    def get_index(self, tokens):
        return [self.index.get(t, 1) for t in tokens]


class OracleIndex:
    def __init__(self, max_tokens=300):
        self.L = lib()
        self.h = C.c_void_p(self.L.fmo_index_create(max_tokens))
        self.vocab = Vocab()

    def close(self):
        if self.h:
            self.L.fmo_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- string level (FuzzyMatch::add_tm(id, Sentence, Tokens)) --
    def add_tm(self, norm, real=None, itoks=None):
        """real=None reproduces add_tm(id, Tokens): the real tokens are the normalised tokens, which the
        oracle models as "no separate real tokens" unless another sentence forces the string path."""
        ids = np.asarray(self.vocab.add_words(norm), dtype=np.uint32)
        n, idx, strs = _itok_args(itoks)
        ts = None if (real is None and not itoks) else "\t".join(real if real is not None else norm).encode("utf-8")
        return self.L.fmo_index_add(self.h, _ptr(ids), len(ids), ts, n, idx, strs)

    def add_ids_bulk(self, off, toks, vocab_size, sfreq=None):
        off = np.ascontiguousarray(off, dtype=np.uint32)
        toks = np.ascontiguousarray(toks, dtype=np.uint32)
        self.L.fmo_index_add_bulk(self.h, _ptr(off), _ptr(toks), len(off) - 1)
        self._bulk_vocab = (vocab_size, sfreq)

    def sort(self, nthreads=None):
        if hasattr(self, "_bulk_vocab"):
            vs, sf = self._bulk_vocab
            sf = None if sf is None else np.ascontiguousarray(sf, dtype=np.uint32)
            self.L.fmo_index_set_vocab(self.h, vs, _ptr(sf) if sf is not None else None)
        else:
            sf = np.asarray(self.vocab.sfreq, dtype=np.uint32)
            self.L.fmo_index_set_vocab(self.h, len(sf), _ptr(sf))
        self.L.fmo_index_sort(self.h, nthreads or effective_cpus())

    def set_arrays(self, suf_sid, suf_pos, buffer, sent_pos, qva, sfreq):
        a = [np.ascontiguousarray(suf_sid, np.uint32), np.ascontiguousarray(suf_pos, np.uint16),
             np.ascontiguousarray(buffer, np.uint32), np.ascontiguousarray(sent_pos, np.uint32),
             np.ascontiguousarray(qva, np.uint32)]
        self.L.fmo_index_set_arrays(self.h, _ptr(a[0]), _ptr(a[1]), len(a[0]), _ptr(a[2]), len(a[2]),
                                    _ptr(a[3]), len(a[3]), _ptr(a[4]), len(a[4]))
        if isinstance(sfreq, (int, np.integer)):   # a vocabulary size: the oracle counts the sentence frequencies itself
            self.L.fmo_index_count_vocab(self.h, int(sfreq))
            return
        sf = np.ascontiguousarray(sfreq, np.uint32)
        self.L.fmo_index_set_vocab(self.h, len(sf), _ptr(sf))

    def set_real(self, s_id, real, itoks=None):
        n, idx, strs = _itok_args(itoks)
        self.L.fmo_index_set_real(self.h, s_id, "\t".join(real).encode("utf-8"), n, idx, strs)

    def set_real_bulk(self, packed, boff, first=0):
        """Real tokens of sentences [first, first + len(boff)) from tab_joined()'s output (no penalty tokens)."""
        packed = np.ascontiguousarray(packed, np.uint8)
        boff = np.ascontiguousarray(boff, np.uint64)
        rc = self.L.fmo_index_set_real_bulk(self.h, first, len(boff), _ptr(packed), _ptr(boff))
        if rc != 0:
            raise RuntimeError("oracle set_real_bulk failed")

    def arrays(self):
        L, h = self.L, self.h
        ns, nb, nsent, nq = (L.fmo_index_num_suffixes(h), L.fmo_index_buffer_size(h),
                             L.fmo_index_num_sentences(h), L.fmo_index_qva_size(h))

        def view(p, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dt).copy()
        return {"suf_sid": view(L.fmo_index_suf_sid(h), ns, np.uint32),
                "suf_pos": view(L.fmo_index_suf_pos(h), ns, np.uint16),
                "buffer": view(L.fmo_index_buffer(h), nb, np.uint32),
                "sent_pos": view(L.fmo_index_sentence_pos(h), nsent, np.uint32),
                "qva": view(L.fmo_index_qva(h), nq, np.uint32)}

    def equal_range(self, ngram, lo=0, hi=0):
        ng = np.ascontiguousarray(ngram, np.uint32)
        out = (C.c_uint64 * 2)()
        self.L.fmo_equal_range(self.h, _ptr(ng), len(ng), lo, hi, C.byref(out), None)
        return int(out[0]), int(out[1])

    def match_ids(self, params, wids, real=None, itoks=None, cap=256, counters=None):
        """FuzzyMatch::match(Sentence real, Tokens pattern, …) on ids.  Returns a structured array."""
        w = np.ascontiguousarray(wids, np.uint32)
        n, idx, strs = _itok_args(itoks)
        ts = None if (real is None and not itoks) else "\t".join(real if real is not None else []).encode("utf-8")
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        r = self.L.fmo_match(self.h, C.byref(params), _ptr(w), len(w), ts, n, idx, strs, _ptr(out), cap,
                             C.byref(counters) if counters is not None else None)
        if r < 0:
            raise RuntimeError("oracle match failed")
        return out[:min(int(r), cap)], int(r)

    def match_ids_prior(self, params, wids, prior, cap=256, prior_sid=None):
        """match() on ids when the caller's vector already holds `prior` matches (lists of token ids); params.nmatch = NEW
        matches wanted (src/fuzzy_match.cc:622-650).  prior_sid: their s_ids (the memo of pair costs is keyed on ids, :637)."""
        w = np.ascontiguousarray(wids, np.uint32)
        p_off = np.zeros(len(prior) + 1, np.uint32)
        p_off[1:] = np.cumsum([len(x) for x in prior])
        p_tok = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in prior]) if len(prior) else np.zeros(1, np.uint32), np.uint32)
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        self.L.fmo_match_prior.restype = C.c_int64
        self.L.fmo_match_prior.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_uint64]
        if prior_sid is not None:
            sid = np.ascontiguousarray(prior_sid, np.uint32)
            assert len(sid) == len(prior)
            self.L.fmo_match_prior_ids.restype = C.c_int64
            self.L.fmo_match_prior_ids.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_uint32, C.c_void_p, C.c_uint64]
            r = self.L.fmo_match_prior_ids(self.h, C.byref(params), _ptr(w), len(w), _ptr(p_off), _ptr(p_tok), _ptr(sid), len(prior), _ptr(out), cap)
        else:
            r = self.L.fmo_match_prior(self.h, C.byref(params), _ptr(w), len(w), _ptr(p_off), _ptr(p_tok), len(prior), _ptr(out), cap)
        if r < 0:
            raise RuntimeError("oracle match failed")
        return out[:min(int(r), cap)], int(r)

    def match_tokens(self, params, tokens, real=None, itoks=None, cap=256):
        """match(const Tokens&, …) / match(Sentence, Tokens, …) with string tokens."""
        return self.match_ids(params, self.vocab.get_index(tokens), real=real, itoks=itoks, cap=cap)

    def subsequence_ids(self, wids, nmatch, no_perfect=False, min_subseq_length=3, min_subseq_ratio=0.3,
                        idf_weighting=False, real=None, itoks=None):
        """FuzzyMatch::subsequence after tokenisation.  Returns (record | None, start of the sub-n-gram)."""
        w = np.ascontiguousarray(wids, np.uint32)
        n, idx, strs = _itok_args(itoks)
        ts = None if (real is None and not itoks) else "\t".join(real if real is not None else []).encode("utf-8")
        out = np.zeros(1, dtype=MATCH_DTYPE)
        pos = C.c_uint32(0)
        r = self.L.fmo_subsequence(self.h, _ptr(w), len(w), ts, n, idx, strs, nmatch, int(no_perfect), min_subseq_length,
                                   min_subseq_ratio, int(idf_weighting), _ptr(out), C.byref(pos))
        if r < 0:
            raise RuntimeError("oracle subsequence failed")
        return (out[0].copy(), int(pos.value)) if r else (None, 0)

    def match_batch(self, params, q_off, q_tok, cap, nthreads=1, want_counters=False, real=None):
        """real: (packed, byte offsets) from tab_joined() — every pattern's `real` Sentence (no penalty tokens)."""
        q_off = np.ascontiguousarray(q_off, np.uint32)
        q_tok = np.ascontiguousarray(q_tok, np.uint32)
        nq = len(q_off) - 1
        out = np.zeros((max(nq, 1), cap), dtype=MATCH_DTYPE)
        cnt = np.zeros(max(nq, 1), dtype=np.uint32)
        ctr = Counters()
        rp = ro = None
        if real is not None:
            rp, ro = np.ascontiguousarray(real[0], np.uint8), np.ascontiguousarray(real[1], np.uint64)
            assert len(ro) == nq
        rc = self.L.fmo_match_batch_real(self.h, C.byref(params), _ptr(q_off), _ptr(q_tok), nq, _ptr(rp) if rp is not None else None,
                                         _ptr(ro) if ro is not None else None, nthreads, _ptr(out),
                                         _ptr(cnt), cap, C.byref(ctr) if want_counters else None)
        if rc != 0:
            raise RuntimeError("oracle batch failed")
        return (out[:nq], cnt[:nq], ctr) if want_counters else (out[:nq], cnt[:nq])
