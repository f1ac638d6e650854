"""Full-size checks (BASELINE config C2: 1 M-sentence corpus, 100 k queries) through size-independent
properties, plus an oracle sample.  Runs in a few seconds on an MI355X (the index is sorted on the GPU)."""
import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(1_000_000, 100_000, 20)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    return ix, c_off, c_tok, q_off, q_tok, vs


def _csr_take(off, tok, idx):
    lens = (off[1:] - off[:-1])[idx]
    new_off = np.zeros(len(idx) + 1, np.uint32)
    new_off[1:] = np.cumsum(lens)
    parts = [tok[off[i]:off[i + 1]] for i in idx]
    return new_off, (np.concatenate(parts) if parts else np.zeros(0, np.uint32)).astype(np.uint32)


def test_suffix_array_is_sorted_and_complete(c2):
    """Every suffix appears once and adjacent suffixes are in SuffixArray::comp order (sampled)."""
    ix, c_off, c_tok, *_ = c2
    sid, pos = ix.array("suf_sid"), ix.array("suf_pos")
    assert len(sid) == len(c_tok)
    key = sid.astype(np.uint64) << np.uint64(16) | pos.astype(np.uint64)
    assert len(np.unique(key)) == len(key)
    buf, sp = ix.array("buffer"), ix.array("sent_pos")
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(sid) - 1, size=20000):
        a = buf[sp[sid[i]] + pos[i]: sp[sid[i]] + 1 + buf[sp[sid[i]]]]
        b = buf[sp[sid[i + 1]] + pos[i + 1]: sp[sid[i + 1]] + 1 + buf[sp[sid[i + 1]]]]
        n = min(len(a), len(b))
        d = np.nonzero(a[:n] != b[:n])[0]
        if len(d):
            assert a[d[0]] < b[d[0]]
        elif len(a) != len(b):
            assert len(a) < len(b)
        else:
            assert sid[i] < sid[i + 1]


def test_indexed_sentences_match_themselves(fmx, c2):
    """A pattern that is an indexed sentence has that sentence (or an identical one) as a score-1 match."""
    ix, c_off, c_tok, *_ = c2
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(c_off) - 1, size=20000)
    q_off, q_tok = _csr_take(c_off, c_tok, idx)
    out, cnt, _ = ix.match_batch(fmx.Params(fuzzy=1.0, nmatch=1), q_off, q_tok, cap=1)
    assert (cnt == 1).all()
    assert (out[:, 0]["score"] == 1.0).all()
    hit = out[:, 0]["s_id"]
    for k in rng.integers(0, len(idx), size=2000):
        a = c_tok[c_off[idx[k]]:c_off[idx[k] + 1]]
        b = c_tok[c_off[hit[k]]:c_off[hit[k] + 1]]
        assert np.array_equal(a, b) and hit[k] <= idx[k]  # ties go to the smaller s_id


def test_batch_permutation_and_sub_batching_invariance(fmx, c2):
    """Results are a function of the pattern alone: shuffling the batch, or cutting it into tiny
    sub-batches with a tiny hit workspace, permutes / leaves the records unchanged."""
    ix, _, _, q_off, q_tok, _ = c2
    prm = fmx.Params(fuzzy=0.7, nmatch=5)
    base_out, base_cnt, _ = ix.match_batch(prm, q_off, q_tok, cap=5)
    again_out, again_cnt, _ = ix.match_batch(prm, q_off, q_tok, cap=5)
    assert np.array_equal(base_cnt, again_cnt) and base_out.tobytes() == again_out.tobytes()  # idempotent
    perm = np.random.default_rng(2).permutation(len(base_cnt))
    p_off, p_tok = _csr_take(q_off, q_tok, perm)
    ix.set_option("sub_queries", 4099)
    try:
        p_out, p_cnt, _ = ix.match_batch(prm, p_off, p_tok, cap=5)
    finally:
        ix.set_option("sub_queries", 1 << 20)
    assert np.array_equal(p_cnt, base_cnt[perm])
    for k in range(len(perm)):
        n = int(p_cnt[k])
        assert p_out[k, :n].tobytes() == base_out[perm[k], :n].tobytes()


def test_threshold_monotonicity_and_order(fmx, c2):
    """Raising the threshold can only remove matches; matches come sorted by (score desc, s_id asc)."""
    ix, _, _, q_off, q_tok, _ = c2
    lo_out, lo_cnt, _ = ix.match_batch(fmx.Params(fuzzy=0.6, nmatch=0), q_off[:20001], q_tok[:int(q_off[20000])], cap=64)
    hi_out, hi_cnt, _ = ix.match_batch(fmx.Params(fuzzy=0.8, nmatch=0), q_off[:20001], q_tok[:int(q_off[20000])], cap=64)
    assert (hi_cnt <= lo_cnt).all()
    for q in range(20000):
        n = int(lo_cnt[q])
        s, i = lo_out[q, :n]["score"], lo_out[q, :n]["s_id"]
        assert (s >= np.float32(0.6)).all()
        assert all((s[k] > s[k + 1]) or (s[k] == s[k + 1] and i[k] < i[k + 1]) for k in range(n - 1))
        m = int(hi_cnt[q])
        hs = set(hi_out[q, :m]["s_id"].tolist())
        assert hs <= set(i.tolist()) or n == 64


def test_oracle_sample_at_scale(fmx, c2):
    """All 100 k queries of config C2 against the oracle on the full 1 M-sentence index, bit for bit, for the
    C3 / C4 / C5 parameter sets (a few seconds of oracle time on the GPU box's host cores each)."""
    ix, c_off, c_tok, q_off, q_tok, vs = c2
    o = oracle_lib.OracleIndex()
    o.set_arrays(ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos"), ix.array("qva"),
                 int(ix.info().vocab_size))   # (the oracle counts the sentence frequencies itself)
    n = len(q_off) - 1
    so, st = q_off[:n + 1], q_tok[:int(q_off[n])]
    for name in ["C3", "C4", "C5"]:
        kw = workloads.PARAM_SETS[name]
        oo, oc = o.match_batch(oracle_lib.Params(**kw), so, st, 5, nthreads=oracle_lib.effective_cpus())
        go, gc, _ = ix.match_batch(fmx.Params(**kw), so, st, cap=5)
        workloads.compare(oo, oc, go, gc, name)


@pytest.fixture(scope="module")
def c3(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(5_000_000, 1_000_000, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    return ix, c_off, c_tok, q_off, q_tok, vs


def test_c3_index_equals_an_independent_oracle_build_at_5m(fmx, c3):
    """The 5 M-sentence index of the headline config, built by the product (suffix sort on the GPU, VocabIndexer::sfreq on the
    host), against the oracle's OWN build from the raw corpus — its restatement of SuffixArray::sort (src/suffix_array.cc:58-102,
    235-251) and of the sentence frequencies (src/vocab_indexer.cc:37-50): every array equal.  The parity tests below feed the
    oracle these arrays; this test is what makes them independent of the product."""
    ix, c_off, c_tok, _, _, vs = c3
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, c_tok, vs)           # no sfreq handed over: the oracle counts its own
    o.sort(oracle_lib.effective_cpus())
    a = o.arrays()
    for k in ["suf_sid", "suf_pos", "buffer", "sent_pos", "qva"]:
        assert np.array_equal(a[k], ix.array(k)), k
    mine = np.asarray(ix.array("sfreq"))
    theirs = np.zeros(len(mine), np.int64)     # counted here, independently: sentences containing each word
    n_sent = len(c_off) - 1
    for s0 in range(0, n_sent, 1 << 20):
        s1 = min(n_sent, s0 + (1 << 20))
        lens = np.diff(c_off[s0:s1 + 1]).astype(np.int64)
        sid = np.repeat(np.arange(s1 - s0, dtype=np.uint64), lens)
        tok = c_tok[int(c_off[s0]):int(c_off[s1])].astype(np.uint64)
        pairs = np.unique(sid << np.uint64(32) | tok)
        theirs += np.bincount((pairs & np.uint64(0xffffffff)).astype(np.int64), minlength=len(theirs))[:len(theirs)]
    n = min(len(mine), len(theirs))
    assert np.array_equal(mine[2:n], theirs[2:n])
    o.close()


def test_c3_headline_config_parity(fmx, gpu_device, c3):
    """The bench workload itself (config C3: 5 M sentences, mean 25 tokens, fuzzy 0.7, nmatch 5): ALL of its 1 M queries, every
    record compared bit for bit with the oracle on the same index (≈ 30 s of oracle time on 16 host cores) — the heavy tail
    (a few dozen queries with 10^4..10^5 survivors, the order-dependent loop of src/fuzzy_match.cc:564-612) is where a sample can
    miss; then the same 1 M queries under C4's and C5's parameters."""
    ix, c_off, c_tok, q_off, q_tok, vs = c3
    kw = workloads.PARAM_SETS["C3"]
    go, gc, _ = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=5)
    o = oracle_lib.OracleIndex()
    o.set_arrays(ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos"), ix.array("qva"),
                 int(ix.info().vocab_size))   # (the oracle counts the sentence frequencies itself)
    n = len(q_off) - 1
    assert n == 1_000_000
    so, st = q_off, q_tok
    oo, oc = o.match_batch(oracle_lib.Params(**kw), so, st, 5, nthreads=oracle_lib.effective_cpus())
    workloads.compare(oo, oc, go, gc, "C3 headline, all 1 M queries")
    del oo, oc
    # all 1 M: results ordered, scores within (fuzzy, 1], ids valid
    n_sent = len(c_off) - 1
    cnt = gc.astype(np.int64)
    assert cnt.max() <= 5 and int((cnt > 0).sum()) > 500_000
    valid = np.arange(5)[None, :] < cnt[:, None]
    sc, sid = go["score"], go["s_id"]
    assert np.all(sid[valid] < n_sent) and np.all(sc[valid] >= np.float32(0.7)) and np.all(sc[valid] <= 1.0)
    both = valid[:, 1:] & valid[:, :-1]
    assert np.all((sc[:, :-1] > sc[:, 1:])[both] | ((sc[:, :-1] == sc[:, 1:]) & (sid[:, :-1] < sid[:, 1:]))[both])
    # BASELINE configs C4 (fuzzy 0.5, --mr 0.5, idf 1) and C5 (contrastive rerank, buffer 32) on the SAME 5 M index:
    # their heavy-tail paths (big candidate sets, list replay, contrast on large accepted sets) see loads the
    # 1 M-sentence corpus does not produce.  All 1 M queries each, every record against the oracle.
    for name in ["C4", "C5"]:
        kw = workloads.PARAM_SETS[name]
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus())
        g2, c2_, _, st2 = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=5, want_stats=True)
        workloads.compare(oo, oc, g2, c2_, name + " on the 5 M index, all 1 M queries")
        assert int(oc.sum()) > 100_000, name
        del oo, oc, g2, c2_
    o.close()
    m = 60_000
    so, st = q_off[:m + 1], q_tok[:int(q_off[m])]
    # the long ranges of this corpus go through the bit planes: make sure that path was what ran
    _, _, _, st3 = ix.match_batch(fmx.Params(**workloads.PARAM_SETS["C3"]), so, st, cap=5, want_stats=True)
    assert st3.n_heavy_hits > st3.n_hits // 2 and st3.n_heavy_pass < st3.n_heavy_hits // 10, (st3.n_hits, st3.n_heavy_hits, st3.n_heavy_pass)
    ix.set_option("heavy_min", 0)  # and the per-hit walk alone gives the same records
    g3, c3_, _ = ix.match_batch(fmx.Params(**workloads.PARAM_SETS["C3"]), so, st, cap=5)
    workloads.compare(go[:m], gc[:m], g3, c3_, "per-hit walk vs bit planes")  # entries beyond the count are unspecified
    ix.set_option("heavy_min", 1024)
    # the one-launch path for micro-batches (fmx_small.hip.hh) on the 5 M index: 20 000 of the queries in calls of 500,
    # against the records of the pipeline (== the oracle's, above); queries with long ranges are handed back to the pipeline
    ix.set_option("small_path", 1)
    n_small = 0
    for lo in range(0, 20_000, 500):
        g4, c4_, _, st4 = ix.match_batch(fmx.Params(**workloads.PARAM_SETS["C3"]), q_off[lo:lo + 501], q_tok, cap=5, want_stats=True)
        workloads.compare(go[lo:lo + 500], gc[lo:lo + 500], g4, c4_, f"one-launch path [{lo}:{lo + 500}]")
        n_small += int(st4.n_small)
    assert 15_000 < n_small <= 20_000, n_small


def test_gpu_sort_equals_independent_oracle_sort_at_1m(fmx, c2):
    """SuffixArray::sort on the GPU against the oracle's own sort (its restatement of src/suffix_array.cc:58-102,235-251)
    on the full 1 M-sentence corpus: every array equal, not a sample."""
    ix, c_off, c_tok, _, _, vs = c2
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, c_tok, vs)
    o.sort(oracle_lib.effective_cpus())
    a = o.arrays()
    for k in ["suf_sid", "suf_pos", "buffer", "sent_pos", "qva"]:
        assert np.array_equal(a[k], ix.array(k)), k


def test_fmi_at_the_size_the_metric_names(fmx, gpu_device, c3, tmp_path_factory, capsys):
    """"5M-sentence .fmi": the headline index through the reference's on-disk format (export_ / import_binarized_fuzzy_matcher,
    src/fuzzy_matcher_binarization.cc:16-51): fmx_index_save writes 5 M Sentence records (real-token strings + an unordered_map
    each), 125 M packed suffixes and the vocabulary with real spellings; fmx_index_load reads it back: every array equal, sentence
    ids and spellings kept, and the loaded index answers the first 20 000 queries of the workload — as ids and as token
    spellings looked up on the device — exactly like the index that was saved."""
    import os
    import time
    import synth
    ix, c_off, c_tok, q_off, q_tok, vs = c3
    words = [b"", b""] + [synth.word(k).encode() for k in range(vs - 2)]
    arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
    offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
    pos = 0
    for k in range(2, len(words)):
        offs[k] = pos
        pos += len(words[k]) + 1
    ix.set_forms_packed(arena, offs)
    path = str(tmp_path_factory.mktemp("fmi") / "c3.fmi")
    t0 = time.time()
    ix.save(path)
    t_save = time.time() - t0
    size = os.path.getsize(path)
    t0 = time.time()
    ld = fmx.Index.load(path)
    t_load = time.time() - t0
    os.unlink(path)
    try:
        for name in ("suf_sid", "suf_pos", "buffer", "sent_pos", "qva", "sfreq"):
            assert np.array_equal(ix.array(name), ld.array(name)), name
        a, b = ix.info(), ld.info()
        assert (a.n_sentences, a.n_suffixes, a.vocab_size, a.max_tokens_in_pattern, a.pt) == (b.n_sentences, b.n_suffixes, b.vocab_size, b.max_tokens_in_pattern, b.pt)
        probe = ["ab", "zz", synth.word(vs - 3), "nosuchword", synth.word(12345)]
        assert list(ld.vocab_lookup(probe)) == list(ix.vocab_lookup(probe))
        for s in (0, 1, 4_999_999):   # a corpus built from ids has no sentence ids of its own: the writer numbers the sentences
            assert ld.sentence_id(s) == str(s)
        t0 = time.time()
        ld.finalize()
        assert not ld.info().real_differs       # the Sentence records spell the normalised tokens: no real-token cost model
        ld.to_device(gpu_device)
        t_up = time.time() - t0
        n = 20_000
        so, st = q_off[:n + 1], q_tok[:int(q_off[n])]
        for name in ["C3", "C5"]:
            kw = workloads.PARAM_SETS[name]
            want = ix.match_batch(fmx.Params(**kw), so, st, cap=5)
            got = ld.match_batch(fmx.Params(**kw), so, st, cap=5)
            workloads.compare(want[0], want[1], got[0], got[1], f"{name} on the index loaded from the .fmi")
        text = synth.pack_text(st)
        want = ix.match_batch(fmx.Params(**workloads.PARAM_SETS["C3"]), so, st, cap=5)
        got = ld.match_batch_text(fmx.Params(**workloads.PARAM_SETS["C3"]), so, text, cap=5)
        workloads.compare(want[0], want[1], got[0], got[1], "text batch on the index loaded from the .fmi")
    finally:
        ld.close()
    with capsys.disabled():
        print(f"\n[.fmi at C3] {size / 1e9:.2f} GB: saved in {t_save:.1f}s, loaded in {t_load:.1f}s, finalized + uploaded in {t_up:.1f}s")
    assert size > 1.5e9
