"""The suffix sorter of the index build on the device (fmx_sort.hip: prefix doubling over a hand-written radix sort) against the
host's sort under the reference's comparator (SuffixArray::sort / comp, src/suffix_array.cc:58-102,235-251): suffix array,
positions and _quickVocabAccess equal on corpora built to tie deep — tiny vocabularies, one token repeated for thousands of
positions, the same sentence hundreds of times, one-token sentences, token id 0, ids beyond the declared vocabulary, ids of up to
26 bits (an index is not practical beyond: _quickVocabAccess has one entry per id) — and the rocPRIM merge sort that rounds 1-3 used (kept for indexes beyond 2^31 suffixes) against both."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _csr(rows):
    off = np.zeros(len(rows) + 1, np.uint32)
    off[1:] = np.cumsum([len(r) for r in rows])
    return off, np.array([t for r in rows for t in r], np.uint32)


def _both(fmx, dev, off, tok, vs):
    a, b = fmx.Index.create(max_tokens=60000), fmx.Index.create(max_tokens=60000)
    a.add_ids(off, tok, vs)
    b.add_ids(off, tok, vs)
    a.sort(dev)
    b.sort(-1, 4)
    for k in ["suf_sid", "suf_pos", "qva", "buffer", "sent_pos"]:
        x, y = a.array(k), b.array(k)
        assert x.shape == y.shape and np.array_equal(x, y), (k, x[:20], y[:20])
    return a


@pytest.mark.parametrize("n_sent,vocab,max_len,seed", [(3000, 2, 40, 1), (2000, 3, 12, 2), (5000, 1, 30, 3), (400, 50, 8, 4),
                                                       (64, 5, 1, 5), (20000, 7, 25, 6), (1, 4, 9, 7), (9000, 2, 3, 8)])
def test_random_corpora_with_deep_ties(fmx, gpu_device, n_sent, vocab, max_len, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len + 1, n_sent)
    off = np.zeros(n_sent + 1, np.uint32)
    off[1:] = np.cumsum(lens)
    tok = rng.integers(0, vocab, int(off[-1])).astype(np.uint32)  # id 0 included
    _both(fmx, gpu_device, off, tok, max(vocab, 2))


def test_long_runs_of_one_token_and_repeated_sentences(fmx, gpu_device):
    rows = [[5] * 3000, [5] * 2999 + [6], [5] * 3000, [6] + [5] * 1500]
    rows += [[2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]] * 700
    rows += [[9]] * 300 + [[9, 9]] * 300
    off, tok = _csr(rows)
    _both(fmx, gpu_device, off, tok, 22)


def test_wide_token_ids(fmx, gpu_device):
    rng = np.random.default_rng(11)
    for top in (0x03FFFFFF, 0x0000FFFF, 70000):
        pool = np.array([0, 1, 2, top, top - 1, top // 2], np.uint64)
        rows = [pool[rng.integers(0, len(pool), rng.integers(1, 9))].astype(np.uint32).tolist() for _ in range(1500)]
        off, tok = _csr(rows)
        _both(fmx, gpu_device, off, tok, 3)  # ids beyond the declared vocabulary: the host's rule grows it


def test_synthetic_corpus(fmx, gpu_device):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import synth
    c_off, c_tok, _, _, vs = synth.corpus_and_queries(60_000, 10, 20)
    _both(fmx, gpu_device, c_off, c_tok, vs)


def test_merge_sort_path_still_sorts(gpu_device, tmp_path):
    """FMX_SORTER=merge is read when the sort starts; a child process keeps the switch out of this one."""
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import importlib\n"
        "fmx = importlib.import_module('fuzzy-match_amd.fmx')\n"
        "rng = np.random.default_rng(5)\n"
        "lens = rng.integers(1, 30, 4000)\n"
        "off = np.zeros(4001, np.uint32); off[1:] = np.cumsum(lens)\n"
        "tok = rng.integers(0, 6, int(off[-1])).astype(np.uint32)\n"
        "a, b = fmx.Index.create(), fmx.Index.create()\n"
        "a.add_ids(off, tok, 6); b.add_ids(off, tok, 6)\n"
        "a.sort(%d); b.sort(-1, 2)\n"
        "assert all(np.array_equal(a.array(k), b.array(k)) for k in ['suf_sid', 'suf_pos', 'qva'])\n"
        "print('merge ok')\n"
    ) % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), gpu_device)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FMX_SORTER="merge", FMX_SORT_TIMING="1"))
    assert r.returncode == 0 and "merge ok" in r.stdout, r.stdout + r.stderr
    assert "merge sort under the comparator" in r.stderr
