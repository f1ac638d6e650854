"""Pins the CPU oracle (oracle/fm_oracle.c) to the reference's own known answers and fixtures.

The reference cannot be built in this image (Boost / OpenNMT Tokenizer absent), so these vectors
are the oracle's pin: test/test.cc known answers (tests/golden/known_answers.json) and the
test-tm1 fixture run against the reference-built index tm1.fmi / tm1.old.fmi.
"""
import os

import numpy as np
import pytest

import fmi_reader
import golden_util
import oracle_lib


def _params(q):
    return oracle_lib.Params(fuzzy=q["fuzzy"], nmatch=q["nmatch"], no_perfect=q["no_perfect"],
                             min_subseq_length=q["ml"], min_subseq_ratio=q["mr"], idf_penalty=q["idf"],
                             costs=tuple(q["costs"]), contrast=q["contrast"], reduce=q["reduce"],
                             contrast_buffer=q["cbuf"])


@pytest.mark.parametrize("case", golden_util.known_answers(), ids=lambda c: c["name"])
def test_known_answers(case):
    ix = oracle_lib.OracleIndex(max_tokens=case["max_tokens"])
    for s in case["sentences"]:
        ix.add_tm(s)
    ix.sort(2)
    for q in case["queries"]:
        res, n = ix.match_tokens(_params(q), q["pattern"])
        if "expect_count" in q:
            assert n == q["expect_count"]
        if "expect_sids" in q:
            assert list(res["s_id"]) == q["expect_sids"]
        if "expect_scores" in q:
            assert np.allclose(res["score"], q["expect_scores"], atol=q["tol"], rtol=0)
        if "exact_scores" in q:
            assert [("%.9g" % s) for s in res["score"]] == q["exact_scores"]
        if "expect_score_minus_penalty" in q:
            assert np.allclose(res["score"] - res["penalty"], q["expect_score_minus_penalty"], atol=q["tol"], rtol=0)


def _oracle_from_fmi(path):
    d = fmi_reader.read_fmi(path)
    assert d["consumed"] == d["size"]
    ix = oracle_lib.OracleIndex(max_tokens=d["max_tokens"])
    ix.set_arrays([s for s, _ in d["suffixes"]], [p for _, p in d["suffixes"]], d["buffer"], d["sent_pos"],
                  d["qva"], d["sfreq"])
    for sid, (ts, itoks) in enumerate(d["real"]):
        ix.set_real(sid, ts.split("\t") if ts else [], itoks)
    ix.vocab.forms = list(d["forms"])
    ix.vocab.index = dict(d["form2index"])
    return ix, d


@pytest.mark.parametrize("fmi", ["tm1.fmi", "tm1.old.fmi"])
def test_tm1_fixture(fmi):
    """test/test.cc:192-196,211-215 (prebuild_tm1, prebuild_old_tm1) through the oracle."""
    ix, d = _oracle_from_fmi(os.path.join(golden_util.GOLDEN, fmi))
    assert d["pt"] == 49 and len(d["ids"]) == 8 and len(d["suffixes"]) == 33
    for case in golden_util.tm1_cases():
        prm = oracle_lib.Params(fuzzy=case["fuzzy"], nmatch=case["nmatch"], no_perfect=case["no_perfect"],
                                min_subseq_length=2, min_subseq_ratio=0)
        res, n = ix.match_tokens(prm, case["norm"], real=case["real"], itoks=case["itoks"])
        golden_util.check_tm1_case(case, list(res["score"]), [d["ids"][s] for s in res["s_id"]])


def test_tm1_exact_scores():
    """Scores the reference itself printed for test-tm1 in the survey probe."""
    ix, d = _oracle_from_fmi(os.path.join(golden_util.GOLDEN, "tm1.fmi"))
    by_id = {c["id"]: c for c in golden_util.tm1_cases()}

    def scores(cid):
        c = by_id[cid]
        prm = oracle_lib.Params(fuzzy=c["fuzzy"], nmatch=c["nmatch"], no_perfect=c["no_perfect"],
                                min_subseq_length=2, min_subseq_ratio=0)
        res, _ = ix.match_tokens(prm, c["norm"], real=c["real"], itoks=c["itoks"])
        return ["%.9g" % s for s in res["score"]]
    assert scores("FUZZY2")[2] == "0.74000001"
    assert scores("CASE2") == ["0.99000001", "0.980000019"]
    assert scores("NUM2") == ["0.980000019"]
    assert scores("NOLIMIT")[-1] == "0.666599989"


def test_rebuilt_index_equals_fixture():
    """Rebuilding tm1 from its tokenised sentences gives the reference's suffix order byte for byte
    (SuffixArray::sort is fully determined by comp, src/suffix_array.cc:235-251)."""
    d = fmi_reader.read_fmi(os.path.join(golden_util.GOLDEN, "tm1.fmi"))
    ix = oracle_lib.OracleIndex()
    for sid in range(len(d["sent_pos"])):
        sp = d["sent_pos"][sid]
        n = d["buffer"][sp]
        ix.add_tm([d["forms"][w] for w in d["buffer"][sp + 1:sp + 1 + n]])
    ix.sort(1)
    a = ix.arrays()
    assert list(zip(a["suf_sid"], a["suf_pos"])) == d["suffixes"]
    assert list(a["buffer"]) == d["buffer"] and list(a["sent_pos"]) == d["sent_pos"]
    assert list(a["qva"]) == d["qva"]
    assert ix.vocab.forms == d["forms"] and ix.vocab.sfreq == d["sfreq"]


def test_edit_distance_char():
    L = oracle_lib.lib()
    assert L.fmo_edit_distance_char(b"", 0, b"abc", 3) == 3
    assert L.fmo_edit_distance_char(b"T", 1, b"", 0) == 1
    assert L.fmo_edit_distance_char(b"kitten", 6, b"sitting", 7) == 3
    assert L.fmo_edit_distance_char(b"T", 1, b"T", 1) == 0


def test_oracle_counts_its_own_sentence_frequencies():
    """VocabIndexer::addWords (src/vocab_indexer.cc:73-90): sfreq[w] = sentences containing w, bumped once per distinct word.  The
    oracle keeps that count itself while sentences are added (round 5: the parity tests no longer hand it the product's counts);
    checked here against a count made with numpy, and through idf-weighted scores (src/fuzzy_match.cc:372-390)."""
    import workloads
    rng = np.random.default_rng(5)
    c_off, c_tok = workloads.dense_corpus(rng, 4000, 60, 1, 14)
    vs = 64
    n_sent = len(c_off) - 1
    sid = np.repeat(np.arange(n_sent, dtype=np.uint64), np.diff(c_off).astype(np.int64))
    pairs = np.unique(sid << np.uint64(32) | c_tok.astype(np.uint64))
    want = np.bincount((pairs & np.uint64(0xffffffff)).astype(np.int64), minlength=vs)[:vs].astype(np.uint32)
    own = workloads.build_oracle(c_off, c_tok, vs, nthreads=2)                 # counts its own
    given = workloads.build_oracle(c_off, c_tok, vs, sfreq=want, nthreads=2)   # handed the numpy count
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, 60, 300)
    prm = oracle_lib.Params(fuzzy=0.4, nmatch=5, idf_penalty=1.0, min_subseq_length=2, min_subseq_ratio=0)
    a, ac = own.match_batch(prm, q_off, q_tok, 5, nthreads=2)
    b, bc = given.match_batch(prm, q_off, q_tok, 5, nthreads=2)
    assert np.array_equal(ac, bc) and ac.sum() > 100
    assert all(a[q, :ac[q]].tobytes() == b[q, :bc[q]].tobytes() for q in range(len(ac)))
    flat = oracle_lib.Params(fuzzy=0.4, nmatch=5, min_subseq_length=2, min_subseq_ratio=0)
    c, cc = own.match_batch(flat, q_off, q_tok, 5, nthreads=2)
    assert any(a[q, :ac[q]].tobytes() != c[q, :cc[q]].tobytes() for q in range(len(ac)))   # the idf term does change scores
    own.close()
    given.close()
