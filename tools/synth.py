"""ctypes wrapper of tools/libfm_synth.so — the portable synthetic corpus / query generator of
SURVEY.md §8d.  Produces CSR (offsets, vocabulary ids) numpy arrays; used by bench.py and tests."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfm_synth.so")
VOCAB_TYPES = 50000
CORPUS_SEED = 12345
QUERY_SEED = 54321
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.fms_corpus_new.restype = C.c_void_p
        L.fms_corpus_new.argtypes = [C.c_uint64, C.c_double, C.c_uint32, C.c_uint64]
        L.fms_queries_new.restype = C.c_void_p
        L.fms_queries_new.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_uint32, C.c_uint64]
        L.fms_free.argtypes = [C.c_void_p]
        for n in ["fms_num_sentences", "fms_num_tokens"]:
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.fms_vocab_size.restype = C.c_uint32
        L.fms_vocab_size.argtypes = [C.c_void_p]
        for n in ["fms_offsets", "fms_tokens"]:
            getattr(L, n).restype = C.c_void_p
            getattr(L, n).argtypes = [C.c_void_p]
        L.fms_word.argtypes = [C.c_uint32, C.c_char_p, C.c_int]
        L.fms_pack_text.restype = C.c_uint64
        L.fms_pack_text.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.fms_is_number.argtypes = [C.c_uint32]
        L.fms_pack_lines_cased.restype = C.c_uint64
        L.fms_pack_lines_cased.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.fms_pack_cased_stream.restype = C.c_uint64
        L.fms_pack_cased_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _take(h):
    L = lib()
    ns, nt = L.fms_num_sentences(h), L.fms_num_tokens(h)
    off = np.frombuffer((C.c_char * ((ns + 1) * 4)).from_address(L.fms_offsets(h)), dtype=np.uint32).copy()
    tok = (np.frombuffer((C.c_char * (nt * 4)).from_address(L.fms_tokens(h)), dtype=np.uint32).copy()
           if nt else np.zeros(0, np.uint32))
    return off, tok


def corpus_and_queries(n_sent, n_queries, mean_len, vocab_types=VOCAB_TYPES, corpus_seed=CORPUS_SEED,
                       query_seed=QUERY_SEED):
    """Returns (c_off, c_tok, q_off, q_tok, vocab_size)."""
    L = lib()
    hc = L.fms_corpus_new(n_sent, float(mean_len), vocab_types, corpus_seed)
    hq = L.fms_queries_new(hc, n_queries, float(mean_len), vocab_types, query_seed)
    try:
        c_off, c_tok = _take(hc)
        q_off, q_tok = _take(hq)
        vs = L.fms_vocab_size(hc)
    finally:
        L.fms_free(hc)
        L.fms_free(hq)
    return c_off, c_tok, q_off, q_tok, vs


def word(k):
    buf = C.create_string_buffer(16)
    lib().fms_word(k, buf, 16)
    return buf.value.decode()


def pack_text(ids, out=None):
    """The patterns as packed token spellings (NUL-terminated, back to back): what a tokenizer writes.  `out`: a uint8 array to
    fill (e.g. pinned memory); None = a new numpy array.  Returns the filled array (a view of `out`)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    L = lib()
    need = L.fms_pack_text(ids.ctypes.data, len(ids), None, 0)
    if out is None:
        out = np.empty(need, np.uint8)
    assert out.nbytes >= need
    L.fms_pack_text(ids.ctypes.data, len(ids), out.ctypes.data, out.nbytes)
    return out[:need]


def is_number(k):
    """Vocabulary type k is written as a number by the cased text (tools/fm_synth.c: fms_is_number)."""
    return bool(lib().fms_is_number(int(k)))


def pack_lines_cased(off, ids, out=None):
    """The sentences as the TEXT the reference CLI would read: one line per sentence, blank-separated, the first word capitalised,
    ~2 % of the tokens numbers.  Returns a uint8 array (a view of `out` when given)."""
    off = np.ascontiguousarray(off, dtype=np.uint32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    L = lib()
    n = len(off) - 1
    need = L.fms_pack_lines_cased(off.ctypes.data, ids.ctypes.data, n, None, 0)
    if out is None:
        out = np.empty(max(need, 1), np.uint8)
    assert out.nbytes >= need
    L.fms_pack_lines_cased(off.ctypes.data, ids.ctypes.data, n, out.ctypes.data, out.nbytes)
    return out[:need]


def pack_cased_stream(off, ids, which, out=None):
    """The NUL-separated spelling stream -p nbr,tag,cas makes of pack_lines_cased's lines: which = 0 normalised tokens (lower-case
    word / the number placeholder), 1 real tokens (case feature "C" / "L" / the digits)."""
    off = np.ascontiguousarray(off, dtype=np.uint32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    L = lib()
    n = len(off) - 1
    need = L.fms_pack_cased_stream(off.ctypes.data, ids.ctypes.data, n, which, None, 0)
    if out is None:
        out = np.empty(max(need, 1), np.uint8)
    assert out.nbytes >= need
    L.fms_pack_cased_stream(off.ctypes.data, ids.ctypes.data, n, which, out.ctypes.data, out.nbytes)
    return out[:need]


NUM_KEY = np.uint32(0xfffffff0)


def cased_vocab_map(c_tok):
    """The vocabulary ids VocabIndexer::addWord (src/vocab_indexer.cc:22-50) hands out when pack_lines_cased's text is indexed under
    -p nbr,tag,cas line by line: ids from 2 in order of first appearance of the NORMALISED token — a word's lower-case spelling, one
    shared placeholder for every number.  Computed here from the generator's ids alone (numpy), independent of any tokenizer.
    Returns (map: generator id -> index id, 1 = not in the corpus; vocab_size)."""
    c_tok = np.asarray(c_tok, dtype=np.uint32)
    k = c_tok - np.uint32(2)
    key = np.where((c_tok >= 2) & (k % np.uint32(40) == 19), NUM_KEY, c_tok)
    uniq, first = np.unique(key, return_index=True)
    order = np.argsort(first, kind="stable")
    top = int(c_tok.max()) + 1 if len(c_tok) else 2
    m = np.ones(max(top, VOCAB_TYPES + 2), np.uint32)
    num_id = 1
    for rank_, u in enumerate(uniq[order]):
        if u == NUM_KEY:
            num_id = rank_ + 2
        else:
            m[u] = rank_ + 2
    ids = np.arange(len(m), dtype=np.uint32)
    isnum = (ids >= 2) & ((ids - np.uint32(2)) % np.uint32(40) == 19)
    m[isnum] = num_id          # every number shares the placeholder's id (1 if the corpus holds no number)
    m[0], m[1] = 0, 1
    return m, len(uniq) + 2
