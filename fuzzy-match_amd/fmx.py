"""ctypes binding of libfmx.so (include/fmx.h) — plumbing for tests and bench.py.

The product is the C-ABI library and the C++ facade; this module only marshals numpy arrays
into it.  It never computes a match itself and there is no fallback: if libfmx.so is missing
or no gfx950 device is present, the calls raise.
"""
import ctypes as C
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the process's first HIP call (see fmx_capi.cc, fmx_runtime_defaults)
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMX_LIB") or os.path.join(_HERE, "libfmx.so")  # FMX_LIB: tuning experiments only

FMX_OK = 0
ARRAYS = {
    "suf_sid": (0, np.uint32), "suf_pos": (1, np.uint16), "buffer": (2, np.uint32),
    "sent_pos": (3, np.uint32), "qva": (4, np.uint32), "sfreq": (5, np.uint32),
    "suf_rec": (6, np.uint64), "idf": (7, np.float32), "real_ids": (8, np.uint32),
    "itok_ids": (9, np.uint32), "itok_dist": (10, np.uint16), "suf_sig": (11, np.uint64), "suf_key": (12, np.dtype([("t", np.uint32, 8)])),
    "gram": (13, np.dtype([("w1", np.uint32), ("w2", np.uint32), ("lo", np.uint32), ("cnt", np.uint32)])),
    "suf_b": (15, np.dtype([("off", np.uint32), ("rem", np.uint16), ("slen", np.uint16), ("prev", np.uint32), ("sid", np.uint32)])),
    # 128-byte buckets of the 5-gram table (fmx_core.hh: G5Bucket): three 40-byte entries + the directory {tags 0, 1 | tag 2, state}
    "gram5": (14, np.dtype([("e", np.dtype([("w", np.uint32, 4), ("lo", np.uint32), ("cnt", np.uint32), ("t", np.uint32, 4)]), 3),
                            ("dir_a", np.uint32), ("dir_b", np.uint32)])),
    "gram7": (16, np.dtype([("w1", np.uint32), ("w2", np.uint32), ("lo", np.uint32), ("cnt", np.uint32)])),
}

MATCH_DTYPE = np.dtype([("score", np.float32), ("penalty", np.float32),
                        ("max_subseq", np.int32), ("s_id", np.uint32)])


class Params(C.Structure):
    """fmx_params — arguments of FuzzyMatch::match (include/fuzzy/fuzzy_match.hh:72-84)."""
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


class Batch(C.Structure):
    _fields_ = [("nq", C.c_uint64), ("q_off", C.c_void_p), ("q_tok", C.c_void_p),
                ("q_real", C.c_void_p), ("q_itok", C.c_void_p), ("max_len", C.c_uint32)]


class TextBatch(C.Structure):
    """fmx_text_batch: patterns as NUL-terminated token spellings back to back."""
    _fields_ = [("nq", C.c_uint64), ("q_off", C.c_void_p), ("text", C.c_void_p), ("n_bytes", C.c_uint64),
                ("real_text", C.c_void_p), ("n_real_bytes", C.c_uint64), ("n_itoks", C.c_uint64), ("itok_slot", C.c_void_p), ("itok_id", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_h2d", C.c_double), ("ms_d2h", C.c_double),
                ("ms_kernel", C.c_double * 16), ("launches", C.c_uint64 * 16),
                ("n_hits", C.c_uint64), ("n_kept", C.c_uint64), ("n_candidates", C.c_uint64),
                ("n_dp", C.c_uint64), ("n_retries", C.c_uint64), ("n_big", C.c_uint64),
                ("n_phase_b", C.c_uint64), ("max_kept", C.c_uint64), ("n_heavy_hits", C.c_uint64),
                ("n_heavy_pass", C.c_uint64), ("pinned", C.c_uint64), ("plane_words", C.c_uint64),
                ("probe_bytes", C.c_uint64), ("probe_lines", C.c_uint64), ("n_light_hits", C.c_uint64), ("n_small", C.c_uint64), ("n_rerank", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [("n_sentences", C.c_uint64), ("n_suffixes", C.c_uint64), ("buffer_size", C.c_uint64),
                ("vocab_size", C.c_uint64), ("max_tokens_in_pattern", C.c_uint64),
                ("pt", C.c_int32), ("sorted", C.c_int32), ("finalized", C.c_int32),
                ("on_device", C.c_int32), ("device", C.c_int32), ("real_differs", C.c_int32),
                ("n_itok", C.c_uint32), ("idf_max", C.c_float), ("device_bytes", C.c_uint64)]


EXPORTS = [
    "fmx_last_error", "fmx_version", "fmx_stage_name", "fmx_device_count", "fmx_index_create",
    "fmx_index_free", "fmx_index_load", "fmx_index_save", "fmx_index_add_tm", "fmx_index_add_ids",
    "fmx_index_sort", "fmx_index_finalize", "fmx_index_to_device", "fmx_vocab_lookup",
    "fmx_index_lookup_real", "fmx_index_intern_itok", "fmx_index_sentence_id", "fmx_index_array",
    "fmx_index_get_info", "fmx_match_batch", "fmx_match_batch_text", "fmx_match_batch_text_keep", "fmx_match_batch_text_begin", "fmx_match_batch_text_end", "fmx_match_batch_lines", "fmx_match_batch_device", "fmx_set_option",
    "fmx_host_alloc", "fmx_host_free", "fmx_index_to_devices", "fmx_set_devices", "fmx_index_reopen", "fmx_match_one", "fmx_match_batch_keep",
    "fmx_subsequence_batch", "fmx_text_tokenize", "fmx_text_free", "fmx_text_num_tokens", "fmx_text_norm", "fmx_text_real",
    "fmx_text_num_itoks", "fmx_text_itok_index", "fmx_text_itok", "fmx_text_num_raw", "fmx_text_raw", "fmx_text_map",
    "fmx_text_detokenize", "fmx_index_add_text", "fmx_index_device_array", "fmx_index_from_arrays", "fmx_text_feature",
    "fmx_match_one_prior", "fmx_match_one_prior_ids", "fmx_match_batch_raw_lines", "fmx_text_tokenize_general", "fmx_index_add_text_bulk", "fmx_match_batch_device_begin", "fmx_match_batch_device_end", "fmx_vocab_lookup_mt", "fmx_index_set_forms",
]
TM_SKIPPED = -100

_lib = None


def build(force=False):
    """Compile libfmx.so in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C fuzzy-match_amd/csrc` "
                           "(there is no Python or CPU fallback for the match path)")
    L = C.CDLL(LIB_PATH)
    L.fmx_last_error.restype = C.c_char_p
    L.fmx_version.restype = C.c_char_p
    L.fmx_stage_name.restype = C.c_char_p
    L.fmx_stage_name.argtypes = [C.c_int]
    L.fmx_index_create.restype = C.c_void_p
    L.fmx_index_create.argtypes = [C.c_int32, C.c_uint64]
    L.fmx_index_free.argtypes = [C.c_void_p]
    L.fmx_index_load.restype = C.c_void_p
    L.fmx_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.fmx_index_save.argtypes = [C.c_void_p, C.c_char_p]
    L.fmx_index_add_tm.restype = C.c_int64
    L.fmx_index_add_tm.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p,
                                   C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
    L.fmx_index_add_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.fmx_index_sort.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fmx_index_finalize.argtypes = [C.c_void_p]
    L.fmx_index_to_device.argtypes = [C.c_void_p, C.c_int]
    L.fmx_vocab_lookup.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint64, C.c_void_p]
    L.fmx_index_lookup_real.restype = C.c_uint32
    L.fmx_index_lookup_real.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.fmx_index_intern_itok.restype = C.c_uint32
    L.fmx_index_intern_itok.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.fmx_index_sentence_id.restype = C.c_char_p
    L.fmx_index_sentence_id.argtypes = [C.c_void_p, C.c_uint32]
    L.fmx_index_array.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.fmx_index_device_array.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.fmx_index_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.fmx_match_batch_text.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(TextBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.POINTER(Stats)]
    L.fmx_match_batch_text_keep.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(TextBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.fmx_match_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_uint32, C.POINTER(Stats)]
    L.fmx_match_batch_keep.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.fmx_match_batch_device.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
    L.fmx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.fmx_match_batch_device_begin.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Batch), C.c_uint64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
    L.fmx_match_batch_device_end.argtypes = [C.c_void_p, C.c_void_p]
    L.fmx_vocab_lookup_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.fmx_index_set_forms.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.fmx_host_alloc.restype = C.c_void_p
    L.fmx_host_alloc.argtypes = [C.c_size_t]
    L.fmx_host_free.argtypes = [C.c_void_p]
    L.fmx_index_to_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.fmx_set_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.fmx_index_reopen.argtypes = [C.c_void_p]
    L.fmx_index_from_arrays.restype = C.c_void_p
    L.fmx_index_from_arrays.argtypes = [C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.fmx_match_one_prior.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fmx_match_one_prior_ids.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fmx_match_one.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fmx_subsequence_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_int32, C.c_int32, C.c_float,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fmx_text_tokenize.restype = C.c_void_p
    L.fmx_text_tokenize.argtypes = [C.c_int32, C.c_char_p, C.c_size_t]
    L.fmx_text_free.argtypes = [C.c_void_p]
    for name in ("num_tokens", "num_itoks", "num_raw"):
        f = getattr(L, "fmx_text_" + name)
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p]
    for name in ("norm", "real", "itok", "raw"):
        f = getattr(L, "fmx_text_" + name)
        f.restype = C.c_char_p
        f.argtypes = [C.c_void_p, C.c_uint32]
    L.fmx_text_itok_index.restype = C.c_uint64
    L.fmx_text_itok_index.argtypes = [C.c_void_p, C.c_uint32]
    L.fmx_text_map.restype = C.c_uint32
    L.fmx_text_map.argtypes = [C.c_void_p, C.c_uint32]
    L.fmx_text_detokenize.restype = C.c_char_p
    L.fmx_text_detokenize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.fmx_index_add_text.restype = C.c_int64
    L.fmx_index_add_text.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    _lib = L
    return L


class FmxError(RuntimeError):
    pass


def _check(rc):
    if rc != FMX_OK:
        raise FmxError(f"fmx error {rc}: {lib().fmx_last_error().decode('utf-8', 'replace')}")


def device_count():
    return lib().fmx_device_count()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = lib().fmx_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise MemoryError("fmx_host_alloc failed")
        self.buf = (C.c_char * max(int(nbytes), 1)).from_address(self.ptr)

    def __del__(self):
        try:
            if self.ptr:
                lib().fmx_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


_pinned_blocks = {}


def pinned_empty(shape, dtype):
    """numpy array over pinned host memory (fmx_host_alloc): fmx_match_batch overlaps its PCIe copies with compute
    when the batch and result buffers are pinned.  The memory lives until pinned_free(array) or interpreter exit."""
    dt = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape)) if shape else 1
    blk = _PinnedBlock(n * dt.itemsize)
    arr = np.frombuffer(blk.buf, dtype=dt, count=n).reshape(shape)
    _pinned_blocks[arr.ctypes.data] = blk
    return arr


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def pinned_free(arr):
    _pinned_blocks.pop(arr.ctypes.data, None)


def tokenize(pt, sentence, general=False):
    """FuzzyMatch::_tokenize_and_normalize through the restricted text front-end (fmx_text_tokenize; general: without the fast path
    ASCII sentences take).  Returns dict(norm, real, itoks {index: str}, raw, map)."""
    L = lib()
    b = sentence.encode("utf-8")
    L.fmx_text_tokenize_general.restype = C.c_void_p
    L.fmx_text_tokenize_general.argtypes = [C.c_int32, C.c_char_p, C.c_size_t]
    h = L.fmx_text_tokenize_general(pt, b, len(b)) if general else L.fmx_text_tokenize(pt, b, len(b))
    if not h:
        raise MemoryError("fmx_text_tokenize failed")
    try:
        n = L.fmx_text_num_tokens(h)
        d = lambda x: x.decode("utf-8", "replace")  # noqa: E731
        return {"norm": [d(L.fmx_text_norm(h, i)) for i in range(n)],
                "real": [d(L.fmx_text_real(h, i)) for i in range(n)],
                "itoks": {int(L.fmx_text_itok_index(h, k)): d(L.fmx_text_itok(h, k)) for k in range(L.fmx_text_num_itoks(h))},
                "raw": [d(L.fmx_text_raw(h, i)) for i in range(L.fmx_text_num_raw(h))],
                "map": [int(L.fmx_text_map(h, k)) for k in range(n + 1)]}
    finally:
        L.fmx_text_free(h)


def detokenize_span(pt, sentence, start, end):
    """Detokenised raw tokens behind the normalised tokens [start, end) of `sentence` (subsequence()'s Match::id suffix)."""
    L = lib()
    b = sentence.encode("utf-8")
    h = L.fmx_text_tokenize(pt, b, len(b))
    try:
        return L.fmx_text_detokenize(h, start, end).decode("utf-8", "replace")
    finally:
        L.fmx_text_free(h)


class Index:
    """Handle on an fmx_index (FuzzyMatch + SuffixArrayIndex of the reference)."""

    def __init__(self, handle):
        if not handle:
            raise FmxError(lib().fmx_last_error().decode("utf-8", "replace"))
        self.h = C.c_void_p(handle)

    @classmethod
    def create(cls, pt=0, max_tokens=300):
        return cls(lib().fmx_index_create(pt, max_tokens))

    @classmethod
    def load(cls, path):
        err = C.create_string_buffer(512)
        h = lib().fmx_index_load(os.fsencode(path), err, 512)
        if not h:
            raise FmxError(err.value.decode("utf-8", "replace"))
        return cls(h)

    @classmethod
    def from_arrays(cls, suf_sid, suf_pos, buffer, sent_pos, qva=None, sfreq=None, vocab_size=0, pt=0, max_tokens=300):
        """fmx_index_from_arrays: adopt an already sorted suffix array + sentence buffer (the reference's SuffixArray members)."""
        suf_sid = np.ascontiguousarray(suf_sid, dtype=np.uint32)
        suf_pos = np.ascontiguousarray(suf_pos, dtype=np.uint16)
        buffer = np.ascontiguousarray(buffer, dtype=np.uint32)
        sent_pos = np.ascontiguousarray(sent_pos, dtype=np.uint32)
        qva = None if qva is None else np.ascontiguousarray(qva, dtype=np.uint32)
        sfreq = None if sfreq is None else np.ascontiguousarray(sfreq, dtype=np.uint32)
        err = C.create_string_buffer(512)
        h = lib().fmx_index_from_arrays(pt, max_tokens, _ptr(suf_sid), _ptr(suf_pos), len(suf_sid), _ptr(buffer), len(buffer),
                                        _ptr(sent_pos), len(sent_pos), _ptr(qva), 0 if qva is None else len(qva), _ptr(sfreq),
                                        vocab_size or (0 if sfreq is None else len(sfreq)), err, 512)
        if not h:
            raise FmxError(err.value.decode("utf-8", "replace"))
        return cls(h)

    def close(self):
        if self.h:
            lib().fmx_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save(self, path):
        _check(lib().fmx_index_save(self.h, os.fsencode(path)))

    def add_tm(self, sid, norm, real=None, itoks=None):
        """add_tm(id, Sentence real, Tokens norm): real = list of real tokens or None; itoks = {idx: str}."""
        arr = (C.c_char_p * len(norm))(*[t.encode("utf-8") for t in norm])
        real_s = None if real is None else "\t".join(real).encode("utf-8")
        itoks = itoks or {}
        n = len(itoks)
        idx = (C.c_uint64 * max(n, 1))(*list(itoks.keys()))
        strs = (C.c_char_p * max(n, 1))(*[v.encode("utf-8") for v in itoks.values()])
        r = lib().fmx_index_add_tm(self.h, sid.encode("utf-8"), arr, len(norm), real_s, n, idx, strs)
        if r == TM_SKIPPED:
            return -1  # ignored like the reference ignores it (empty / too long)
        if r < 0:
            _check(int(r))
        return int(r)

    def add_text(self, sid, sentence):
        """add_tm(id, sentence): tokenised by the restricted text front-end; -1 = empty segment, not indexed."""
        b = sentence.encode("utf-8")
        r = lib().fmx_index_add_text(self.h, sid.encode("utf-8"), b, len(b))
        if r == TM_SKIPPED:
            return -1
        if r < 0:
            _check(int(r))
        return int(r)

    def add_text_bulk(self, sids, sentences):
        """fmx_index_add_text_bulk: n sentences (str) with their ids (str); returns the sentence id per line (-1 = not indexed)."""
        n = len(sentences)
        bs = [s.encode("utf-8") for s in sentences]
        ids = (C.c_char_p * max(n, 1))(*[s.encode("utf-8") for s in sids])
        ptr = (C.c_char_p * max(n, 1))(*bs)
        lens = (C.c_size_t * max(n, 1))(*[len(b) for b in bs])
        out = (C.c_int64 * max(n, 1))()
        L = lib()
        L.fmx_index_add_text_bulk.restype = C.c_int64
        L.fmx_index_add_text_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        r = L.fmx_index_add_text_bulk(self.h, ids, ptr, lens, n, out)
        if r < 0:
            _check(int(r))
        return [(-1 if out[i] == TM_SKIPPED else int(out[i])) for i in range(n)]

    def add_ids(self, off, toks, vocab_size):
        off = np.ascontiguousarray(off, dtype=np.uint32)
        toks = np.ascontiguousarray(toks, dtype=np.uint32)
        _check(lib().fmx_index_add_ids(self.h, _ptr(off), _ptr(toks), len(off) - 1, vocab_size))

    def sort(self, device=-1, nthreads=None):
        _check(lib().fmx_index_sort(self.h, device, nthreads or os.cpu_count() or 1))

    def finalize(self):
        _check(lib().fmx_index_finalize(self.h))

    def to_device(self, device=0):
        _check(lib().fmx_index_to_device(self.h, device))

    def to_devices(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().fmx_index_to_devices(self.h, arr, len(devices)))

    def reopen(self):
        _check(lib().fmx_index_reopen(self.h))

    def match_one(self, params, wids, q_real=None, q_itok=None, cap=None):
        """FuzzyMatch::match for one pattern through the request coalescer (thread-safe, blocking)."""
        wids = np.ascontiguousarray(wids, dtype=np.uint32)
        cap = cap or (params.nmatch if params.nmatch > 0 else 64)
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        cnt, tot = C.c_uint32(0), C.c_uint32(0)
        if q_real is not None:
            q_real = np.ascontiguousarray(q_real, dtype=np.uint32)
        if q_itok is not None:
            q_itok = np.ascontiguousarray(q_itok, dtype=np.uint32)
        _check(lib().fmx_match_one(self.h, C.byref(params), _ptr(wids), len(wids), _ptr(q_real), _ptr(q_itok), _ptr(out),
                                   cap, C.byref(cnt), C.byref(tot)))
        return out[:cnt.value], tot.value

    def match_one_prior(self, params, wids, prior, q_real=None, q_itok=None, cap=None, prior_sid=None):
        """match() for one pattern when the caller's vector already holds `prior` matches (lists of token ids) and the
        contrastive factor is > 0; params.nmatch = NEW matches wanted.  prior_sid: their Match::s_id (fmx_match_one_prior_ids: the
        reference's memo of pair costs is keyed on ids)."""
        wids = np.ascontiguousarray(wids, dtype=np.uint32)
        cap = cap or (params.nmatch if params.nmatch > 0 else 64)
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        cnt, tot = C.c_uint32(0), C.c_uint32(0)
        p_off = np.zeros(len(prior) + 1, np.uint32)
        p_off[1:] = np.cumsum([len(x) for x in prior])
        p_tok = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in prior]) if len(prior) else np.zeros(1, np.uint32), dtype=np.uint32)
        if q_real is not None:
            q_real = np.ascontiguousarray(q_real, dtype=np.uint32)
        if q_itok is not None:
            q_itok = np.ascontiguousarray(q_itok, dtype=np.uint32)
        if prior_sid is not None:
            sid = np.ascontiguousarray(prior_sid, dtype=np.uint32)
            assert len(sid) == len(prior)
            _check(lib().fmx_match_one_prior_ids(self.h, C.byref(params), _ptr(wids), len(wids), _ptr(q_real), _ptr(q_itok), _ptr(p_off),
                                                 _ptr(p_tok), _ptr(sid), len(prior), _ptr(out), cap, C.byref(cnt), C.byref(tot)))
        else:
            _check(lib().fmx_match_one_prior(self.h, C.byref(params), _ptr(wids), len(wids), _ptr(q_real), _ptr(q_itok), _ptr(p_off),
                                             _ptr(p_tok), len(prior), _ptr(out), cap, C.byref(cnt), C.byref(tot)))
        return out[:cnt.value], tot.value

    def info(self):
        i = Info()
        _check(lib().fmx_index_get_info(self.h, C.byref(i)))
        return i

    def array(self, name):
        which, dt = ARRAYS[name]
        p = C.c_void_p()
        n = C.c_uint64()
        _check(lib().fmx_index_array(self.h, which, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def device_array(self, name, replica=0):
        """A device-built accelerator array copied out of HBM (tests)."""
        which, dt = ARRAYS[name]
        n = C.c_uint64()
        _check(lib().fmx_index_device_array(self.h, replica, which, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=dt)
        if n.value:
            _check(lib().fmx_index_device_array(self.h, replica, which, _ptr(out), out.nbytes, C.byref(n)))
        return out

    def vocab_lookup(self, tokens):
        arr = (C.c_char_p * len(tokens))(*[t.encode("utf-8") for t in tokens])
        ids = np.zeros(len(tokens), dtype=np.uint32)
        _check(lib().fmx_vocab_lookup(self.h, arr, len(tokens), _ptr(ids)))
        return ids

    def lookup_real(self, s):
        b = s.encode("utf-8")
        return lib().fmx_index_lookup_real(self.h, b, len(b))

    def intern_itok(self, s):
        b = s.encode("utf-8")
        return lib().fmx_index_intern_itok(self.h, b, len(b))

    def sentence_id(self, s_id):
        return lib().fmx_index_sentence_id(self.h, s_id).decode("utf-8")

    def set_option(self, name, value):
        _check(lib().fmx_set_option(self.h, name.encode(), int(value)))

    def match_batch(self, params, q_off, q_tok, q_real=None, q_itok=None, cap=None, want_stats=False, out=None, keep=None):
        """FuzzyMatch::match over a CSR batch of vocabulary-id patterns (host buffers).
        Returns (records[nq, cap], counts[nq], totals[nq]) (+ Stats).  `out` = (records, counts, totals) arrays to
        fill (e.g. pinned_empty ones) instead of fresh pageable arrays."""
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        q_tok = np.ascontiguousarray(q_tok, dtype=np.uint32)
        nq = len(q_off) - 1
        if cap is None:
            cap = params.nmatch if params.nmatch > 0 else 64
        if out is not None:
            out, cnt, tot = out
            assert out.shape[0] >= nq and out.shape[1] == cap and out.dtype == MATCH_DTYPE
        else:
            out = np.zeros((max(nq, 1), cap), dtype=MATCH_DTYPE)
            cnt = np.zeros(max(nq, 1), dtype=np.uint32)
            tot = np.zeros(max(nq, 1), dtype=np.uint32)
        if q_real is not None:
            q_real = np.ascontiguousarray(q_real, dtype=np.uint32)
        if q_itok is not None:
            q_itok = np.ascontiguousarray(q_itok, dtype=np.uint32)
        b = Batch(nq, _ptr(q_off), _ptr(q_tok), _ptr(q_real), _ptr(q_itok), 0)
        st = Stats()
        if keep is not None:  # (device pointer of records, device pointer of counts): a copy stays in HBM
            _check(lib().fmx_match_batch_keep(self.h, C.byref(params), C.byref(b), _ptr(out), _ptr(cnt), _ptr(tot), cap,
                                              keep[0], keep[1], C.byref(st)))
        else:
            _check(lib().fmx_match_batch(self.h, C.byref(params), C.byref(b), _ptr(out), _ptr(cnt), _ptr(tot), cap,
                                         C.byref(st)))
        res = (out[:nq], cnt[:nq], tot[:nq])
        return res + (st,) if want_stats else res

    def match_batch_text(self, params, q_off, text, cap=None, want_stats=False, out=None, keep=None, real_text=None, itoks=None):
        """fmx_match_batch_text: the patterns as token spellings (uint8 array: NUL-terminated spellings back to back, q_off in
        tokens); the vocabulary ids are looked up on the device."""
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        nq = len(q_off) - 1
        if cap is None:
            cap = params.nmatch if params.nmatch > 0 else 64
        if out is not None:
            out, cnt, tot = out
            assert out.shape[0] >= nq and out.shape[1] == cap and out.dtype == MATCH_DTYPE
        else:
            out = np.zeros((max(nq, 1), cap), dtype=MATCH_DTYPE)
            cnt = np.zeros(max(nq, 1), dtype=np.uint32)
            tot = np.zeros(max(nq, 1), dtype=np.uint32)
        b = TextBatch(nq, _ptr(q_off), _ptr(text), len(text), None, 0, 0, None, None)
        if real_text is not None:   # the real tokens as a second stream of spellings
            real_text = np.ascontiguousarray(real_text, dtype=np.uint8)
            b.real_text, b.n_real_bytes = _ptr(real_text), len(real_text)
        if itoks is not None:       # (slots, interned ids) of the batch's penalty tokens
            i_slot = np.ascontiguousarray(itoks[0], dtype=np.uint64)
            i_id = np.ascontiguousarray(itoks[1], dtype=np.uint32)
            b.n_itoks, b.itok_slot, b.itok_id = len(i_slot), _ptr(i_slot), _ptr(i_id)
        st = Stats()
        if keep is not None:  # (device pointers: records, counts) also left in HBM for a collective that follows
            _check(lib().fmx_match_batch_text_keep(self.h, C.byref(params), C.byref(b), _ptr(out), _ptr(cnt), _ptr(tot), cap,
                                                   C.c_void_p(keep[0]), C.c_void_p(keep[1]), C.byref(st)))
        else:
            _check(lib().fmx_match_batch_text(self.h, C.byref(params), C.byref(b), _ptr(out), _ptr(cnt), _ptr(tot), cap, C.byref(st)))
        res = (out[:nq], cnt[:nq], tot[:nq])
        return res + (st,) if want_stats else res

    def match_batch_lines(self, params, text, n_lines, cap=None, want_stats=False):
        """fmx_match_batch_lines: pre-tokenised lines (bytes / uint8 array: blank-separated tokens, every line ended by a line feed)."""
        text = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        if cap is None:
            cap = params.nmatch if params.nmatch > 0 else 64
        out = np.zeros((max(n_lines, 1), cap), dtype=MATCH_DTYPE)
        cnt = np.zeros(max(n_lines, 1), dtype=np.uint32)
        tot = np.zeros(max(n_lines, 1), dtype=np.uint32)
        st = Stats()
        lib().fmx_match_batch_lines.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_uint32, C.POINTER(Stats)]
        _check(lib().fmx_match_batch_lines(self.h, C.byref(params), _ptr(text), len(text), n_lines, _ptr(out), _ptr(cnt), _ptr(tot), cap, C.byref(st)))
        res = (out[:n_lines], cnt[:n_lines], tot[:n_lines])
        return res + (st,) if want_stats else res

    def match_batch_raw_lines(self, params, text, n_lines, cap=None, want_stats=False):
        """fmx_match_batch_raw_lines: raw UTF-8 lines (bytes / uint8 array, every line ended by a line feed) — match(const std::string&, ...)."""
        text = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        if cap is None:
            cap = params.nmatch if params.nmatch > 0 else 64
        out = np.zeros((max(n_lines, 1), cap), dtype=MATCH_DTYPE)
        cnt = np.zeros(max(n_lines, 1), dtype=np.uint32)
        tot = np.zeros(max(n_lines, 1), dtype=np.uint32)
        st = Stats()
        lib().fmx_match_batch_raw_lines.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_uint32, C.POINTER(Stats)]
        _check(lib().fmx_match_batch_raw_lines(self.h, C.byref(params), _ptr(text), len(text), n_lines, _ptr(out), _ptr(cnt), _ptr(tot), cap, C.byref(st)))
        res = (out[:n_lines], cnt[:n_lines], tot[:n_lines])
        return res + (st,) if want_stats else res

    def subsequence_batch(self, q_off, q_tok, nmatch, no_perfect=False, min_subseq_length=3, min_subseq_ratio=0.3,
                          idf_weighting=False, q_real=None, q_itok=None):
        """FuzzyMatch::subsequence over a CSR batch of vocabulary-id patterns.
        Returns (record[nq], found[nq], position[nq])."""
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        q_tok = np.ascontiguousarray(q_tok, dtype=np.uint32)
        nq = len(q_off) - 1
        out = np.zeros(max(nq, 1), dtype=MATCH_DTYPE)
        found = np.zeros(max(nq, 1), dtype=np.uint32)
        pos = np.zeros(max(nq, 1), dtype=np.uint32)
        if q_real is not None:
            q_real = np.ascontiguousarray(q_real, dtype=np.uint32)
        if q_itok is not None:
            q_itok = np.ascontiguousarray(q_itok, dtype=np.uint32)
        b = Batch(nq, _ptr(q_off), _ptr(q_tok), _ptr(q_real), _ptr(q_itok), 0)
        _check(lib().fmx_subsequence_batch(self.h, C.byref(b), nmatch, int(no_perfect), min_subseq_length,
                                           min_subseq_ratio, int(idf_weighting), _ptr(out), _ptr(found), _ptr(pos)))
        return out[:nq], found[:nq], pos[:nq]

    def match_batch_device(self, params, nq, d_q_off, d_q_tok, total_tokens, d_out, d_count, d_total, cap,
                           stream=0, d_q_real=0, d_q_itok=0, max_len=0, want_stats=True):
        """Same with raw device pointers (ints) — buffers owned by the caller (e.g. torch tensors).  want_stats=False: no
        per-stage timing, passes back to back, concurrent calls (own stream each) chained on the GPU."""
        b = Batch(nq, d_q_off, d_q_tok, d_q_real or None, d_q_itok or None, max_len)
        st = Stats() if want_stats else None
        _check(lib().fmx_match_batch_device(self.h, C.byref(params), C.byref(b), total_tokens, d_out, d_count,
                                            d_total or None, cap, stream or None, C.byref(st) if want_stats else None))
        return st


def _device_begin(self, params, nq, d_q_off, d_q_tok, total_tokens, d_out, d_count, d_total, cap, stream=0, d_q_real=0,
                  d_q_itok=0, max_len=0):
    """fmx_match_batch_device_begin: enqueue the call, return a handle for device_end (same thread)."""
    b = Batch(nq, d_q_off, d_q_tok, d_q_real or None, d_q_itok or None, max_len)
    h = C.c_void_p()
    _check(lib().fmx_match_batch_device_begin(self.h, C.byref(params), C.byref(b), total_tokens, d_out, d_count, d_total or None, cap,
                                              stream or None, C.byref(h)))
    return h


def _device_end(self, handle):
    _check(lib().fmx_match_batch_device_end(self.h, handle))


def _set_forms_packed(self, arena, offsets):
    """fmx_index_set_forms from a packed buffer: arena = NUL-terminated spellings (numpy uint8), offsets[k] = where id k's starts
    (uint64; ~0 = none).  Returns the pointer array (keep it alive with the arena while lookups by pointer are made)."""
    base = arena.ctypes.data
    ptrs = np.where(offsets == np.uint64(0xffffffffffffffff), np.uint64(0), offsets + np.uint64(base)).astype(np.uint64)
    _check(lib().fmx_index_set_forms(self.h, _ptr(ptrs), len(ptrs)))
    return ptrs


def _vocab_lookup_ptrs(self, ptrs, out, nthreads):
    """fmx_vocab_lookup_mt over an array of char pointers (numpy uint64) into `out` (numpy uint32, e.g. pinned)."""
    _check(lib().fmx_vocab_lookup_mt(self.h, _ptr(ptrs), len(ptrs), _ptr(out), nthreads))


def _text_begin(self, params, q_off, text, out, cap, keep=None, real_text=None):
    """fmx_match_batch_text_begin: q_off / text / out = (records, counts, totals) are pinned numpy arrays the caller keeps alive
    until text_end (real_text: the real tokens as a second stream of spellings, likewise).  Returns the handle."""
    nq = len(q_off) - 1
    b = TextBatch(nq, _ptr(q_off), _ptr(text), len(text), None, 0, 0, None, None)
    if real_text is not None:
        b.real_text, b.n_real_bytes = _ptr(real_text), len(real_text)
    L = lib()
    L.fmx_match_batch_text_begin.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(TextBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    _check(L.fmx_match_batch_text_begin(self.h, C.byref(params), C.byref(b), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), cap,
                                        C.c_void_p(keep[0]) if keep else None, C.c_void_p(keep[1]) if keep else None, C.byref(h)))
    return h


def _text_end(self, handle):
    L = lib()
    L.fmx_match_batch_text_end.argtypes = [C.c_void_p, C.c_void_p]
    _check(L.fmx_match_batch_text_end(self.h, handle))


Index.text_begin = _text_begin
Index.text_end = _text_end
Index.set_forms_packed = _set_forms_packed
Index.vocab_lookup_ptrs = _vocab_lookup_ptrs
Index.device_begin = _device_begin
Index.device_end = _device_end


def stage_names():
    names = []
    for s in range(16):
        n = lib().fmx_stage_name(s).decode()
        if not n:
            break
        names.append(n)
    return names
