"""FuzzyMatch::subsequence (src/fuzzy_match.cc:250-365).

The reference holds no test or fixture for subsequence() (test/test.cc never calls it), so the oracle's restatement
(fmo_subsequence) is pinned by HAND-DERIVED cases only — each expected value below is worked out in the comment
from the reference's code — and the GPU path (fmx_subsequence_batch: k_sub_extent / k_sub_eval) is compared with
the oracle bit for bit on randomised corpora.
"""
import numpy as np
import pytest

import oracle_lib
import test_gpu_realtok as rt
import workloads


def _oracle(sents):
    o = oracle_lib.OracleIndex()
    for s in sents:
        o.add_tm(s.split())
    o.sort(1)
    return o


def test_hand_derived_longest_common_ngram():
    # ids by first appearance: a=2 b=3 c=4 d=5 e=6 x=7 y=8.  Pattern "q b c d z" (q, z unknown -> weight -1 ends the
    # inner loop, :284), ml = 3: the only queued sub-n-gram is (weight 3, position 1, length 3) = "b c d".
    # Its suffix range in suffix order: s2 "b c d" (end of sentence sorts first), s0 "b c d e", s1 "b c d y".
    #   s2: s = 3, p = 5, diff_word = 100 / 5 = 20, two inserts = 40 -> best = s2, score int(10000 - 4000) / 10000 = 0.6,
    #       max_distance = 40
    #   s0: two replacements = 40, not < 40; candidates = {s2, s0} reaches number_of_matches = 2: stop
    o = _oracle(["a b c d e", "x b c d y", "b c d"])
    rec, pos = o.subsequence_ids(o.vocab.get_index("q b c d z".split()), 2, min_subseq_length=3, min_subseq_ratio=0)
    assert rec is not None and (int(rec["s_id"]), int(rec["max_subseq"]), pos) == (2, 3, 1)
    assert rec["score"] == np.float32(0.6)
    # number_of_matches = 0: candidates.size() < 0 never holds, nothing is scored (:312)
    assert o.subsequence_ids(o.vocab.get_index("q b c d z".split()), 0, min_subseq_length=3, min_subseq_ratio=0)[0] is None
    # pattern shorter than min_subseq_length: false (:270-271)
    assert o.subsequence_ids(o.vocab.get_index("b c".split()), 2, min_subseq_length=3, min_subseq_ratio=0)[0] is None
    # the ratio raises the minimum: int(0.9 * 5) = 4 > 3, no sub-n-gram of 4 known words in a row
    assert o.subsequence_ids(o.vocab.get_index("q b c d z".split()), 2, min_subseq_length=3, min_subseq_ratio=0.9)[0] is None


def test_hand_derived_perfect_and_no_perfect():
    o = _oracle(["a b c d e", "x b c d y", "b c d"])
    pat = o.vocab.get_index("b c d".split())
    # s2 is visited first and costs 0: best = s2, score 1, max_distance = 0, break (:349)
    rec, pos = o.subsequence_ids(pat, 3, no_perfect=False, min_subseq_length=3, min_subseq_ratio=0)
    assert (int(rec["s_id"]), float(rec["score"]), int(rec["max_subseq"]), pos) == (2, 1.0, 3, 0)
    # no_perfect: s2 goes to `perfect` (:329-332); s0: s = 5, p = 3, diff_word = 20, two deletions = 40 -> best = s0, 0.6;
    # s1 costs 40 too, not < 40
    rec, pos = o.subsequence_ids(pat, 3, no_perfect=True, min_subseq_length=3, min_subseq_ratio=0)
    assert (int(rec["s_id"]), int(rec["max_subseq"]), pos) == (0, 3, 0) and rec["score"] == np.float32(0.6)


def test_hand_derived_idf_weighting_and_position_order():
    # N = 4; a and b occur in 3 sentences (idf log(4/3) = 0.2877), x y z in one (idf log 4 = 1.3863); u is unknown.
    # Pattern "a b a b x y z u u u", p = 10, diff_word = 100 / 10 = 10 against every sentence (all shorter).
    #  by length: the longest sub-n-grams (7, 6, 5 tokens) span "b x", which occurs nowhere: empty ranges, popped;
    #    of the four-token ones position 0 pops first (operator<: equal weights, larger position is smaller, :245):
    #    "a b a b" = s0, six inserts = 60 -> score 0.4, max_subseq 4, position 0
    #  by idf: every non-empty sub-n-gram heavier than "x y z" (4.159) would have to contain it plus more tokens,
    #    i.e. span "b x": empty.  "x y z" = s1, seven inserts = 70 -> score 0.3, max_subseq 3, position 4
    o = _oracle(["a b a b", "x y z", "a b c", "b a d"])
    pat = o.vocab.get_index("a b a b x y z u u u".split())
    rec, pos = o.subsequence_ids(pat, 1, min_subseq_length=3, min_subseq_ratio=0, idf_weighting=False)
    assert (int(rec["s_id"]), int(rec["max_subseq"]), pos) == (0, 4, 0) and rec["score"] == np.float32(0.4)
    rec, pos = o.subsequence_ids(pat, 1, min_subseq_length=3, min_subseq_ratio=0, idf_weighting=True)
    assert (int(rec["s_id"]), int(rec["max_subseq"]), pos) == (1, 3, 4) and rec["score"] == np.float32(0.3)


def _check_batch(ix, o, q_off, q_tok, kw, real=None, itoks=None, q_real=None, q_itok=None, tag=""):
    rec, found, pos = ix.subsequence_batch(q_off, q_tok, q_real=q_real, q_itok=q_itok, **kw)
    n_found = 0
    for q in range(len(q_off) - 1):
        ref, rpos = o.subsequence_ids(q_tok[q_off[q]:q_off[q + 1]], real=None if real is None else real[q],
                                      itoks=None if itoks is None else itoks[q], **kw)
        if ref is None:
            assert found[q] == 0, (tag, kw, q)
        else:
            n_found += 1
            assert found[q] == 1 and rec[q].tobytes() == ref.tobytes() and pos[q] == rpos, (tag, kw, q, ref, rec[q], rpos, pos[q])
    return n_found


SUB_PARAMS = [dict(nmatch=1), dict(nmatch=5, no_perfect=True), dict(nmatch=3, idf_weighting=True),
              dict(nmatch=8, min_subseq_length=2, min_subseq_ratio=0.5, no_perfect=True, idf_weighting=True),
              dict(nmatch=2, min_subseq_length=5, min_subseq_ratio=0.0), dict(nmatch=0)]


@pytest.mark.gpu
def test_gpu_subsequence_synthetic_bit_exact(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 500, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, vs)
    total = 0
    for kw in SUB_PARAMS:
        total += _check_batch(ix, o, q_off, q_tok, kw, tag="synthetic")
    assert total > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_gpu_subsequence_dense_corpus_bit_exact(fmx, gpu_device, seed):
    """Tiny vocabulary: long ranges, ties between sub-n-grams, perfect matches, many candidates per range."""
    rng = np.random.default_rng(seed)
    vocab = 6
    lens = rng.integers(1, 14, 1500)
    c_off = np.zeros(len(lens) + 1, np.uint32)
    c_off[1:] = np.cumsum(lens)
    c_tok = rng.integers(2, vocab + 2, int(c_off[-1])).astype(np.uint32)
    qs = []
    for _ in range(300):
        if rng.random() < 0.5:
            s = int(rng.integers(0, len(lens)))
            t = c_tok[c_off[s]:c_off[s + 1]].copy()
        else:
            t = rng.integers(2, vocab + 2, int(rng.integers(1, 16))).astype(np.uint32)
        if rng.random() < 0.3 and len(t):
            t[int(rng.integers(0, len(t)))] = 1  # ｟unk｠ splits the sub-n-grams
        qs.append(t)
    q_off = np.zeros(len(qs) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(t) for t in qs])
    q_tok = np.concatenate(qs).astype(np.uint32)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(2)
    total = 0
    for kw in SUB_PARAMS:
        total += _check_batch(ix, o, q_off, q_tok, kw, tag=f"dense{seed}")
    assert total > 800


@pytest.mark.gpu
def test_gpu_subsequence_real_tokens_and_itoks(fmx, gpu_device):
    """The full cost model (real tokens, case features, penalty strings) inside subsequence's edit distance."""
    rng = np.random.default_rng(21)
    ix = fmx.Index.create(pt=49)
    o = oracle_lib.OracleIndex()
    sents = [rt.make_sentence(rng, 2, 10) for _ in range(300)]
    for i, (norm, real, itoks) in enumerate(sents):
        assert ix.add_tm(str(i), norm, real=real, itoks=itoks) == i
        o.add_tm(norm, real=real, itoks=itoks)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o.sort(2)
    queries = []
    for _ in range(200):
        norm, real, _ = sents[int(rng.integers(0, len(sents)))]
        norm, real = rt.edit(rng, norm, real)
        queries.append((norm, real, rt.rand_itoks(rng, len(norm))))
    q_off = np.zeros(len(queries) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(q[0]) for q in queries])
    q_tok = np.concatenate([ix.vocab_lookup(q[0]) for q in queries]).astype(np.uint32)
    q_real = np.asarray([ix.lookup_real(r) for q in queries for r in q[1]], np.uint32)
    q_itok = np.zeros(int(q_off[-1]) + len(queries), np.uint32)
    for qi, (norm, real, itoks) in enumerate(queries):
        for k, v in itoks.items():
            q_itok[int(q_off[qi]) + qi + k] = ix.intern_itok(v)
    total = 0
    for kw in [dict(nmatch=4, min_subseq_length=2, min_subseq_ratio=0), dict(nmatch=6, min_subseq_length=2, min_subseq_ratio=0.3, no_perfect=True),
               dict(nmatch=2, min_subseq_length=1, min_subseq_ratio=0, idf_weighting=True)]:
        total += _check_batch(ix, o, q_off, q_tok, kw, real=[q[1] for q in queries], itoks=[q[2] for q in queries],
                              q_real=q_real, q_itok=q_itok, tag="real")
    assert total > 300


@pytest.mark.gpu
def test_gpu_subsequence_skips_thousands_of_perfect_matches(fmx, gpu_device):
    """no_perfect with 3000 indexed copies of the sub-n-gram's sentence: every copy goes to `perfect`
    (src/fuzzy_match.cc:329-332) before the first real candidate is scored — more than the 1024 entries the seen list starts
    with; the chunk is re-run with a longer list (the reference has no limit)."""
    rng = np.random.default_rng(9)
    base = [2, 3, 4, 5, 6]
    sents = [base] * 3000 + [[2, 3, 4, 5, 7], [2, 3, 4, 9, 6, 8], [3, 4, 5, 6]] + [list(rng.integers(2, 30, size=6)) for _ in range(200)]
    c_off = np.zeros(len(sents) + 1, np.uint32)
    c_off[1:] = np.cumsum([len(x) for x in sents])
    c_tok = np.asarray([t for x in sents for t in x], np.uint32)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, 32)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, 32)
    qs = [base, [2, 3, 4, 5], [9, 2, 3, 4, 5, 6, 9], [3, 4, 5, 6]]
    q_off = np.zeros(len(qs) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(x) for x in qs])
    q_tok = np.asarray([t for x in qs for t in x], np.uint32)
    n = 0
    for kw in [dict(nmatch=3, no_perfect=True, min_subseq_length=3, min_subseq_ratio=0), dict(nmatch=1, no_perfect=True, min_subseq_length=2, min_subseq_ratio=0),
               dict(nmatch=5, no_perfect=False, min_subseq_length=3, min_subseq_ratio=0)]:
        n += _check_batch(ix, o, q_off, q_tok, kw, tag="perfect")
    assert n >= 6
    o.close()


@pytest.mark.gpu
def test_gpu_subsequence_patterns_beyond_the_lds_rows(fmx, gpu_device):
    """Patterns of 650-900 tokens (--max-tokens-in-pattern above the default): the DP rows of k_sub_eval no longer fit the LDS
    and live in HBM (k_sub_eval<true>) — same records as the oracle, next to short patterns in the same batch."""
    rng = np.random.default_rng(5)
    vocab = 40
    lens = np.concatenate([rng.integers(600, 1000, 60), rng.integers(3, 30, 400)])
    rng.shuffle(lens)
    c_off = np.zeros(len(lens) + 1, np.uint32)
    c_off[1:] = np.cumsum(lens)
    c_tok = rng.integers(2, vocab + 2, int(c_off[-1])).astype(np.uint32)
    long_ids = np.flatnonzero(lens >= 650)
    qs = []
    for s in long_ids[:12]:
        t = c_tok[c_off[s]:c_off[s + 1]].copy()
        for _ in range(int(rng.integers(0, 12))):  # a few substitutions; one pattern stays a perfect match
            t[int(rng.integers(0, len(t)))] = int(rng.integers(2, vocab + 2))
        qs.append(t[:900])
    qs += [rng.integers(2, vocab + 2, int(rng.integers(2, 20))).astype(np.uint32) for _ in range(20)]
    q_off = np.zeros(len(qs) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(t) for t in qs])
    q_tok = np.concatenate(qs).astype(np.uint32)
    assert int(np.diff(q_off).max()) > 650
    ix = fmx.Index.create(max_tokens=1000)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = oracle_lib.OracleIndex(max_tokens=1000)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(2)
    total = 0
    for kw in (dict(nmatch=1), dict(nmatch=3, no_perfect=True), dict(nmatch=2, idf_weighting=True)):
        total += _check_batch(ix, o, q_off, q_tok, kw, tag="long patterns")
    assert total > 30
