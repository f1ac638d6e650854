"""CPU-side check of the HIP decomposition: the serial emulation built from the kernels' own per-lane
functions (tests/emul) must reproduce the oracle bit for bit.  Also checks the product's CPU suffix
sort against the oracle's restatement of SuffixArray::sort."""
import numpy as np
import pytest

import emul_lib
import oracle_lib
import workloads


@pytest.fixture(scope="module")
def synth_pair(fmx):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(10000, 1000, 25)  # BASELINE config C1
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(-1, 4)
    ix.finalize()
    o = workloads.build_oracle(c_off, c_tok, vs, nthreads=4)
    return ix, o, q_off, q_tok


def test_cpu_builder_matches_reference_order(synth_pair):
    ix, o, _, _ = synth_pair
    a = o.arrays()
    for k in ["suf_sid", "suf_pos", "buffer", "sent_pos", "qva"]:
        assert np.array_equal(a[k], ix.array(k)), k


@pytest.mark.parametrize("name", list(workloads.PARAM_SETS))
def test_emul_synthetic(fmx, synth_pair, name):
    ix, o, q_off, q_tok = synth_pair
    kw = workloads.PARAM_SETS[name]
    cap = kw.get("nmatch", 5) or 64
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=4)
    eo, ec, _ = emul_lib.Emul(ix).match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
    workloads.compare(oo, oc, eo, ec, name)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_emul_dense_edge_cases(fmx, seed):
    rng = np.random.default_rng(seed)
    vocab = [3, 6, 12][seed - 1]
    c_off, c_tok = workloads.dense_corpus(rng, 400, vocab)
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, vocab, 400)
    ix = fmx.Index.create(max_tokens=20)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(-1, 2)
    ix.finalize()
    o = oracle_lib.OracleIndex(max_tokens=20)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(2)
    em = emul_lib.Emul(ix)
    for name, kw in workloads.PARAM_SETS.items():
        cap = kw.get("nmatch", 5) or 512
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=2)
        eo, ec, _ = em.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        workloads.compare(oo, oc, eo, ec, f"{name}/seed{seed}")


def test_emul_tm1_fixture_full_cost_model(fmx):
    """The reference's test-tm1 fixture through the emulated pipeline: real-token, case, number and
    penalty-token costs (edit_distance<true>), with the interned ids the C ABI hands to the kernels."""
    import os
    import fmi_reader
    import golden_util
    path = os.path.join(golden_util.GOLDEN, "tm1.fmi")
    ix = fmx.Index.load(path)
    ix.finalize()
    d = fmi_reader.read_fmi(path)
    for case in golden_util.tm1_cases():
        wids = ix.vocab_lookup(case["norm"])
        p = len(wids)
        q_real = np.asarray([ix.lookup_real(r) for r in case["real"]], np.uint32)
        q_itok = np.zeros(p + 1, np.uint32)
        for k, v in case["itoks"].items():
            q_itok[k] = ix.intern_itok(v)
        prm = fmx.Params(fuzzy=case["fuzzy"], nmatch=case["nmatch"], no_perfect=case["no_perfect"],
                         min_subseq_length=2, min_subseq_ratio=0)
        cap = case["nmatch"] or 16
        eo, ec, _ = emul_lib.Emul(ix).match_batch(prm, np.asarray([0, p], np.uint32), wids, q_real=q_real, q_itok=q_itok,
                                                  cap=cap)
        n = int(ec[0])
        golden_util.check_tm1_case(case, list(eo[0, :n]["score"]), [d["ids"][s] for s in eo[0, :n]["s_id"]])


@pytest.mark.parametrize("seed,vocab,n_sent", [(1, 3, 3000), (2, 6, 3000), (3, 40, 4000), (4, 400, 6000)])
def test_table_ranges_equal_the_references_equal_range(fmx, seed, vocab, n_sent):
    """The probe's range function (trigram table, 5-gram buckets with the tokens a group shares, 7-gram entries of the heavy
    5-grams, keys, token stream) against the oracle's restatement of SuffixArray::equal_range (src/suffix_array.cc:105-212) for
    n-grams of every length 1..16: n-grams cut out of the corpus (present), the same with one token changed at every position
    (mostly absent, or present elsewhere), and random ones.  Tiny vocabularies make groups of thousands of suffixes (buckets that
    overflow, deep 5-grams), the larger one singletons."""
    rng = np.random.default_rng(seed)
    c_off, c_tok = workloads.dense_corpus(rng, n_sent, vocab, min_len=1, max_len=24)
    ix = fmx.Index.create(max_tokens=30)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(-1, 4)
    ix.finalize()
    o = oracle_lib.OracleIndex(max_tokens=30)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(4)
    em = emul_lib.Emul(ix)
    assert len(em.a["gram5"]) > 0
    stride = 16
    rows, lens = [], []
    for _ in range(4000):
        s = int(rng.integers(0, len(c_off) - 1))
        a, b = int(c_off[s]), int(c_off[s + 1])
        if b - a < 1:
            continue
        st = int(rng.integers(a, b))
        m = int(rng.integers(1, min(stride, b - st) + 1))
        ng = np.zeros(stride, np.uint32)
        ng[:m] = c_tok[st:st + m]
        rows.append(ng.copy()); lens.append(m)
        k = int(rng.integers(0, m))                       # one token changed
        ng2 = ng.copy()
        ng2[k] = rng.integers(2, vocab + 4)               # (vocab + 2, vocab + 3: words the index does not hold)
        rows.append(ng2); lens.append(m)
        m3 = int(rng.integers(1, stride + 1))             # random n-gram
        ng3 = np.zeros(stride, np.uint32)
        ng3[:m3] = rng.integers(2, vocab + 2, size=m3)
        rows.append(ng3); lens.append(m3)
    rows, lens = np.asarray(rows, np.uint32), np.asarray(lens, np.uint32)
    got = em.ngram_ranges(rows, lens)
    n_hit = n_deep = 0
    for k in range(len(lens)):
        lo, hi = o.equal_range(rows[k, :lens[k]])
        g_lo, g_hi = int(got[k, 0]), int(got[k, 1])
        if hi > lo:
            assert (g_lo, g_hi) == (lo, hi), (k, lens[k], rows[k], (lo, hi), (g_lo, g_hi))
            n_hit += 1
            n_deep += int(lens[k] >= 7 and hi - lo >= 1)
        else:
            assert g_hi == g_lo, (k, lens[k], rows[k], (lo, hi), (g_lo, g_hi))
    assert n_hit > 3000 and n_deep > 100
    o.close()
