"""GPU parity tests proper: the HIP path, called through the C ABI (libfmx.so), against the CPU oracle
on the same seeded inputs.  Bit-exact: identical match ids, score bits, penalties, max_subseq, order."""
import os

import numpy as np
import pytest

import fmi_reader
import golden_util
import oracle_lib
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1(fmx, gpu_device):
    """BASELINE config C1: 10k-sentence synthetic corpus, 1k queries; index sorted on the GPU."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(10000, 1000, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)           # GPU suffix sort
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, vs)
    return ix, o, q_off, q_tok


def test_gpu_suffix_sort_matches_reference_order(c1):
    ix, o, _, _ = c1
    a = o.arrays()
    for k in ["suf_sid", "suf_pos", "buffer", "sent_pos", "qva"]:
        assert np.array_equal(a[k], ix.array(k)), k


@pytest.mark.parametrize("name", list(workloads.PARAM_SETS))
def test_c1_param_sets(fmx, c1, name):
    ix, o, q_off, q_tok = c1
    kw = workloads.PARAM_SETS[name]
    cap = kw.get("nmatch", 5) or 64
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
    go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
    workloads.compare(oo, oc, go, gc, name)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_dense_edge_cases(fmx, gpu_device, seed):
    rng = np.random.default_rng(seed)
    vocab = [3, 6, 12][seed - 1]
    c_off, c_tok = workloads.dense_corpus(rng, 400, vocab)
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, vocab, 400)
    ix = fmx.Index.create(max_tokens=20)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = oracle_lib.OracleIndex(max_tokens=20)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(2)
    a = o.arrays()
    assert np.array_equal(a["suf_sid"], ix.array("suf_sid")) and np.array_equal(a["suf_pos"], ix.array("suf_pos"))
    for name, kw in workloads.PARAM_SETS.items():
        cap = kw.get("nmatch", 5) or 512
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=2)
        go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        workloads.compare(oo, oc, go, gc, f"{name}/seed{seed}")


def test_sub_batching_and_workspace_growth(fmx, c1):
    """Tiny sub-batches and a deliberately undersized hit workspace must not change results."""
    ix, o, q_off, q_tok = c1
    kw = workloads.PARAM_SETS["wide_all"]
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 64, nthreads=oracle_lib.effective_cpus())
    ix.set_option("sub_queries", 37)
    try:
        go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=64, want_stats=True)
    finally:
        ix.set_option("sub_queries", 1 << 20)
    workloads.compare(oo, oc, go, gc, "wide_all/sub37")


@pytest.mark.parametrize("fmi", ["tm1.fmi", "tm1.old.fmi"])
def test_tm1_fixture_on_gpu(fmx, gpu_device, fmi):
    """The reference's own fixture (test/data/test-tm1 against tm1.fmi): real-token, case, number and
    penalty-token costs through the FULL edit-distance kernel."""
    path = os.path.join(golden_util.GOLDEN, fmi)
    ix = fmx.Index.load(path)
    ix.finalize()
    ix.to_device(gpu_device)
    d = fmi_reader.read_fmi(path)
    for case in golden_util.tm1_cases():
        wids = ix.vocab_lookup(case["norm"])
        p = len(wids)
        q_off = np.asarray([0, p], np.uint32)
        q_real = np.asarray([ix.lookup_real(r) for r in case["real"]], np.uint32)
        q_itok = np.zeros(p + 1, np.uint32)
        for k, v in case["itoks"].items():
            q_itok[k] = ix.intern_itok(v)
        prm = fmx.Params(fuzzy=case["fuzzy"], nmatch=case["nmatch"], no_perfect=case["no_perfect"],
                         min_subseq_length=2, min_subseq_ratio=0)
        cap = case["nmatch"] or 16
        go, gc, gt = ix.match_batch(prm, q_off, wids, q_real=q_real, q_itok=q_itok, cap=cap)
        n = int(gc[0])
        golden_util.check_tm1_case(case, list(go[0, :n]["score"]), [d["ids"][s] for s in go[0, :n]["s_id"]])


def test_known_answers_on_gpu(fmx, gpu_device):
    """test/test.cc known answers (tests/golden/known_answers.json) through the string-level add_tm +
    vocab_lookup + match path of the C ABI."""
    for case in golden_util.known_answers():
        ix = fmx.Index.create(max_tokens=case["max_tokens"])
        for i, s in enumerate(case["sentences"]):
            ix.add_tm(str(i), s)
        ix.sort(gpu_device)
        ix.finalize()
        ix.to_device(gpu_device)
        for q in case["queries"]:
            wids = ix.vocab_lookup(q["pattern"])
            prm = fmx.Params(fuzzy=q["fuzzy"], nmatch=q["nmatch"], no_perfect=q["no_perfect"],
                             min_subseq_length=q["ml"], min_subseq_ratio=q["mr"], idf_penalty=q["idf"],
                             costs=tuple(q["costs"]), contrast=q["contrast"], reduce=q["reduce"],
                             contrast_buffer=q["cbuf"])
            go, gc, gt = ix.match_batch(prm, np.asarray([0, len(wids)], np.uint32), wids, cap=16)
            res = go[0, :int(gc[0])]
            if "expect_count" in q:
                assert len(res) == q["expect_count"], case["name"]
            if "expect_sids" in q:
                assert list(res["s_id"]) == q["expect_sids"], case["name"]
            if "expect_scores" in q:
                assert np.allclose(res["score"], q["expect_scores"], atol=q["tol"], rtol=0), case["name"]
            if "exact_scores" in q:
                assert [("%.9g" % s) for s in res["score"]] == q["exact_scores"], case["name"]
            if "expect_score_minus_penalty" in q:
                assert np.allclose(res["score"] - res["penalty"], q["expect_score_minus_penalty"], atol=q["tol"],
                                   rtol=0), case["name"]


def test_big_candidate_sets(fmx, gpu_device):
    """Thousands of surviving candidates per query: the 1024-thread radix sorter (k_cand_big), multi-chunk
    replay and the global-scratch select path, plus s_max / workspace growth retries."""
    rng = np.random.default_rng(7)
    vocab = 8
    c_off, c_tok = workloads.dense_corpus(rng, 30000, vocab, min_len=4, max_len=10)
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, vocab, 48)
    ix = fmx.Index.create(max_tokens=40)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = oracle_lib.OracleIndex(max_tokens=40)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(oracle_lib.effective_cpus())
    for name, kw in [("all", dict(fuzzy=0.2, nmatch=0, min_subseq_length=2, min_subseq_ratio=0)),
                     ("top10_idf", dict(fuzzy=0.0, nmatch=10, min_subseq_length=2, min_subseq_ratio=0, idf_penalty=0.7)),
                     ("contrast", dict(fuzzy=0.3, nmatch=6, min_subseq_length=2, min_subseq_ratio=0, contrast=0.4,
                                       contrast_buffer=0))]:
        cap = 128
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
        go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap, want_stats=True)
        workloads.compare(oo, oc, go, gc, name)
        assert st.n_big > 0 and st.max_kept > 1024, (name, st.n_big, st.max_kept)


def test_many_passes_in_one_asynchronous_group_with_a_small_workspace(fmx, gpu_device):
    """fmx_match_batch enqueues up to 32 passes before it reads any of their counters.  Several passes of one group
    then report that the queues / segments / per-query capacity they ran with were too small: each report has to be
    judged against the limits the pass was enqueued with, not against limits an earlier report has already raised
    (a stale comparison used to end in 'a query queued more hits than it has')."""
    rng = np.random.default_rng(17)
    vocab = 8
    c_off, c_tok = workloads.dense_corpus(rng, 8000, vocab, min_len=4, max_len=10)
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, vocab, 12)
    ix = fmx.Index.create(max_tokens=40)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    ix.set_option("initial_segment", 1024)
    ix.set_option("host_sub_queries", 2)   # 6 passes, one asynchronous group
    o = oracle_lib.OracleIndex(max_tokens=40)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(oracle_lib.effective_cpus())
    kw = dict(fuzzy=0.2, nmatch=0, min_subseq_length=2, min_subseq_ratio=0)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 128, nthreads=oracle_lib.effective_cpus())
    go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=128, want_stats=True)
    workloads.compare(oo, oc, go, gc, "small workspace, 6 passes")
    assert st.n_retries > 1, st.n_retries
    go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=128, want_stats=True)  # grown: no retry left
    workloads.compare(oo, oc, go, gc, "second call")
    assert st.n_retries == 0, st.n_retries


def test_long_sentences_beyond_threshold_table(fmx, gpu_device):
    """Sentences longer than phase A's 1024-entry length table (decided by phase B instead) and patterns
    longer than 256 tokens (signature planes overflow; DP rows through LDS)."""
    long_hit = False
    rng = np.random.default_rng(11)
    vocab = 40
    lens = np.concatenate([rng.integers(1000, 1600, 40), rng.integers(200, 400, 40), rng.integers(3, 30, 200)])
    rng.shuffle(lens)
    c_off = np.zeros(len(lens) + 1, np.uint32)
    c_off[1:] = np.cumsum(lens)
    c_tok = rng.integers(2, vocab + 2, int(c_off[-1])).astype(np.uint32)
    # queries: noisy copies of whole sentences and of long slices
    qs = []
    for s in rng.integers(0, len(lens), 120):
        t = c_tok[c_off[s]:c_off[s + 1]].copy()
        if len(t) > 50 and rng.random() < 0.5:
            a = int(rng.integers(0, len(t) // 3))
            t = t[a:a + int(rng.integers(len(t) // 2, len(t) - a + 1))]
        t = t[:550]  # documented limit of the DP kernels: ~590 pattern tokens (LDS rows)
        n_edit = int(rng.integers(0, max(1, len(t) // 10) + 1))
        t[rng.integers(0, len(t), n_edit)] = rng.integers(2, vocab + 2, n_edit)
        qs.append(t)
    q_off = np.zeros(len(qs) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(t) for t in qs])
    q_tok = np.concatenate(qs).astype(np.uint32)
    ix = fmx.Index.create(max_tokens=2000)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = oracle_lib.OracleIndex(max_tokens=2000)
    o.add_ids_bulk(c_off, c_tok, vocab + 2)
    o.sort(oracle_lib.effective_cpus())
    for name, kw in [("f05", dict(fuzzy=0.5, nmatch=5)), ("f02_all", dict(fuzzy=0.2, nmatch=0, min_subseq_ratio=0.05)),
                     ("idf", dict(fuzzy=0.4, nmatch=8, idf_penalty=0.5)),
                     ("costs", dict(fuzzy=0.3, nmatch=5, costs=(0.2, 0.2, 0.7)))]:
        cap = 64
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
        go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        workloads.compare(oo, oc, go, gc, name)
        assert int(np.sum(oc)) > 0, name
        for q in range(len(qs)):
            long_hit |= bool(np.any(lens[oo["s_id"][q, :oc[q]]] > 1024))
    assert long_hit  # at least one accepted match is a sentence outside the table

