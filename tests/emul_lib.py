"""ctypes wrapper of tests/emul/libfmx_emul.so — TEST-ONLY serial emulation of the HIP pipeline built
from the same per-lane functions the kernels run (fuzzy-match_amd/csrc/fmx_core.hh)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul")
LIB_PATH = os.path.join(_HERE, "libfmx_emul.so")
MATCH_DTYPE = np.dtype([("score", np.float32), ("penalty", np.float32),
                        ("max_subseq", np.int32), ("s_id", np.uint32)])


class CoreParams(C.Structure):
    """fmx::Params (fmx_core.hh)."""
    _fields_ = [("fuzzy", C.c_float), ("nmatch", C.c_uint32), ("no_perfect", C.c_int32),
                ("min_subseq_length", C.c_int32), ("min_subseq_ratio", C.c_float), ("idf_penalty", C.c_float),
                ("ins", C.c_float), ("del_", C.c_float), ("rep", C.c_float), ("contrast", C.c_float),
                ("reduce", C.c_int32), ("cbuf", C.c_int32), ("idf_max", C.c_float), ("max_tokens", C.c_uint32)]


class EmulIndex(C.Structure):
    _fields_ = [("suf", C.c_void_p), ("suf_sig", C.c_void_p), ("suf_key", C.c_void_p), ("gram", C.c_void_p), ("gram5", C.c_void_p), ("gram7", C.c_void_p), ("suf_b", C.c_void_p), ("tok", C.c_void_p), ("sent_pos", C.c_void_p),
                ("qva", C.c_void_p), ("idf", C.c_void_p), ("real", C.c_void_p), ("itok", C.c_void_p),
                ("itok_dist", C.c_void_p), ("n_itok", C.c_uint32), ("n_qva", C.c_uint32), ("gram_mask", C.c_uint32), ("gram5_buckets", C.c_uint32), ("gram7_mask", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(LIB_PATH)
        _lib.emul_match_batch.argtypes = [C.POINTER(EmulIndex), C.POINTER(CoreParams), C.c_uint64] + [C.c_void_p] * 7 + \
                                         [C.c_uint32, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Emul:
    """Runs the emulated pipeline over the derived arrays of a finalized fmx index."""

    def __init__(self, fmx_index):
        self.ix = fmx_index
        info = fmx_index.info()
        self.info = info
        self.a = {n: fmx_index.array(n) for n in ["suf_rec", "suf_sig", "suf_key", "gram", "gram5", "gram7", "suf_b", "buffer", "sent_pos", "qva", "idf"]}
        self.a["real_ids"] = fmx_index.array("real_ids") if info.real_differs else None
        self.a["itok_ids"] = fmx_index.array("itok_ids") if info.real_differs else None

    def _index(self):
        info = self.ix.info()
        return EmulIndex(_p(self.a["suf_rec"]), _p(self.a["suf_sig"]), _p(self.a["suf_key"]), _p(self.a["gram"]), _p(self.a["gram5"]), _p(self.a["gram7"]), _p(self.a["suf_b"]), _p(self.a["buffer"]), _p(self.a["sent_pos"]),
                         _p(self.a["qva"]), _p(self.a["idf"]), None, None, None, info.n_itok, len(self.a["qva"]),
                         max(len(self.a["gram"]), 1) - 1, len(self.a["gram5"]), max(len(self.a["gram7"]), 1) - 1)

    def ngram_ranges(self, ngrams, lens):
        """mgram_range_hashed for every row of `ngrams` (n x stride uint32; lens[k] tokens of row k count): n x 2 (lo, hi)."""
        ngrams = np.ascontiguousarray(ngrams, np.uint32)
        lens = np.ascontiguousarray(lens, np.uint32)
        out = np.zeros((len(lens), 2), np.uint32)
        e = self._index()
        L = lib()
        L.emul_ngram_ranges.argtypes = [C.POINTER(EmulIndex), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.emul_ngram_ranges.restype = None
        L.emul_ngram_ranges(C.byref(e), _p(ngrams), _p(lens), len(lens), ngrams.shape[1], out.ctypes.data_as(C.c_void_p))
        return out

    def match_batch(self, p, q_off, q_tok, q_real=None, q_itok=None, cap=None):
        """p: the fmx Params (fmx_params layout)."""
        info = self.ix.info()
        dist = self.ix.array("itok_dist") if info.n_itok else None
        e = EmulIndex(_p(self.a["suf_rec"]), _p(self.a["suf_sig"]), _p(self.a["suf_key"]), _p(self.a["gram"]), _p(self.a["gram5"]), _p(self.a["gram7"]), _p(self.a["suf_b"]), _p(self.a["buffer"]), _p(self.a["sent_pos"]),
                      _p(self.a["qva"]), _p(self.a["idf"]),
                      _p(self.a["real_ids"]) if self.a["real_ids"] is not None else None,
                      _p(self.a["itok_ids"]) if self.a["itok_ids"] is not None else None,
                      _p(dist) if dist is not None else None, info.n_itok, len(self.a["qva"]),
                      max(len(self.a["gram"]), 1) - 1, len(self.a["gram5"]), max(len(self.a["gram7"]), 1) - 1)
        cp = CoreParams(p.fuzzy, p.nmatch, p.no_perfect, p.min_subseq_length, p.min_subseq_ratio, p.idf_penalty,
                        p.insert_cost, p.delete_cost, p.replace_cost, p.contrast, p.reduce,
                        p.nmatch if p.contrast_buffer == -1 else p.contrast_buffer,
                        info.idf_max if p.idf_penalty != 0 else 0.01, min(info.max_tokens_in_pattern, 0xfffffffe))
        q_off = np.ascontiguousarray(q_off, np.uint32)
        q_tok = np.ascontiguousarray(q_tok, np.uint32)
        nq = len(q_off) - 1
        if cap is None:
            cap = p.nmatch if p.nmatch > 0 else 64
        out = np.zeros((max(nq, 1), cap), dtype=MATCH_DTYPE)
        cnt = np.zeros(max(nq, 1), np.uint32)
        tot = np.zeros(max(nq, 1), np.uint32)
        stats = np.zeros(8, np.uint64)
        qr = np.ascontiguousarray(q_real, np.uint32) if q_real is not None else None
        qi = np.ascontiguousarray(q_itok, np.uint32) if q_itok is not None else None
        rc = lib().emul_match_batch(C.byref(e), C.byref(cp), nq, _p(q_off), _p(q_tok) if len(q_tok) else None,
                               _p(qr) if qr is not None else None, _p(qi) if qi is not None else None,
                               _p(out), _p(cnt), _p(tot), cap, _p(stats))
        assert rc == 0, rc
        self.stats = stats
        return out[:nq], cnt[:nq], tot[:nq]
