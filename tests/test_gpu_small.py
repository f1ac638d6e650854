"""The one-launch path for micro-batches (fmx_small.hip.hh: probe, hit walk, candidates, edit distance, replay and top-N of
one query per wave in a single kernel) against the oracle, bit for bit — on its own and through fmx_match_one / the
coalescer — including the queries it hands back to the pipeline (long patterns, long ranges, many hits or survivors) and
the parameter sets it does not take at all (src/fuzzy_match.cc:435-681)."""
import threading

import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu


def _index(fmx, gpu_device, c_off, c_tok, vs, **kw):
    ix = fmx.Index.create(**kw)
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    ix.set_option("small_path", 1)
    return ix


def _check_in_slices(fmx, ix, o, q_off, q_tok, kw, cap, step, label):
    """Returns (queries answered by the one-launch kernel, queries in all)."""
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
    nq = len(q_off) - 1
    n_small = 0
    for lo in range(0, nq, step):
        hi = min(nq, lo + step)
        go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off[lo:hi + 1], q_tok, cap=cap, want_stats=True)  # absolute offsets into q_tok
        workloads.compare(oo[lo:hi], oc[lo:hi], go, gc, f"{label} [{lo}:{hi}]")
        n_small += int(st.n_small)
    return n_small, nq


def test_small_path_synthetic(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(40000, 1500, 25)
    ix = _index(fmx, gpu_device, c_off, c_tok, vs)
    o = workloads.build_oracle(c_off, c_tok, vs)
    for name in ["C1", "C3", "no_perfect", "fuzzy_zero", "ml_negative", "wide_all", "C4", "C5", "costs_101"]:  # the last four: partly / not eligible
        kw = workloads.PARAM_SETS[name]
        n_small, nq = _check_in_slices(fmx, ix, o, q_off, q_tok, kw, kw["nmatch"] or 64, 200, name)
        if name in ("C1", "C3", "no_perfect", "fuzzy_zero", "ml_negative"):
            assert n_small > 0.8 * nq, (name, n_small, nq)   # nearly every query is answered by the one kernel
        elif name != "wide_all":
            assert n_small == 0, (name, n_small)             # idf / contrast / non-unit costs: the pipeline
    # one query per call, and an nmatch below / above what is found
    kw = dict(fuzzy=0.5, nmatch=2)
    _check_in_slices(fmx, ix, o, q_off[:41], q_tok, kw, 2, 1, "singles")
    kw = dict(fuzzy=0.3, nmatch=40, contrast_buffer=40)
    _check_in_slices(fmx, ix, o, q_off[:301], q_tok, kw, 40, 150, "nmatch 40")


def test_small_path_dense_and_edges(fmx, gpu_device):
    """Tiny vocabulary: long ranges, hundreds of hits and survivors per query (handed back to the pipeline), UNK / empty /
    over-long / id-0 patterns (workloads.edge_queries)."""
    rng = np.random.default_rng(91)
    c_off, c_tok = workloads.dense_corpus(rng, 3000, 7, min_len=1, max_len=14)
    q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, 7, 500)
    ix = _index(fmx, gpu_device, c_off, c_tok, 9, max_tokens=24)
    o = oracle_lib.OracleIndex(max_tokens=24)
    o.add_ids_bulk(c_off, c_tok, 9)
    o.sort(2)
    for kw in [dict(fuzzy=0.6, nmatch=5), dict(fuzzy=0.0, nmatch=10, min_subseq_length=4, min_subseq_ratio=0.2),
               dict(fuzzy=0.4, nmatch=3, min_subseq_length=1, min_subseq_ratio=0), dict(fuzzy=0.8, nmatch=1, no_perfect=True),
               dict(fuzzy=0.5, nmatch=60, contrast_buffer=60, min_subseq_length=2, min_subseq_ratio=0)]:
        n_small, nq = _check_in_slices(fmx, ix, o, q_off, q_tok, kw, kw["nmatch"], 125, str(kw))
        assert 0 < n_small < nq, (kw, n_small, nq)  # some queries answered here, some handed back to the pipeline


def test_small_path_long_sentences_and_patterns(fmx, gpu_device):
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(70, 300, 60), rng.integers(3, 40, 600)])
    rng.shuffle(lens)
    c_off = np.zeros(len(lens) + 1, np.uint32)
    c_off[1:] = np.cumsum(lens)
    c_tok = rng.integers(2, 300, size=int(c_off[-1])).astype(np.uint32)
    offs, toks = [0], []
    for i in range(200):  # copies of indexed sentences with a few edits: patterns of 3..300 tokens
        s = int(rng.integers(0, len(lens)))
        t = list(c_tok[c_off[s]:c_off[s + 1]])
        for _ in range(int(rng.integers(0, 3))):
            t[int(rng.integers(0, len(t)))] = int(rng.integers(2, 300))
        toks += t
        offs.append(len(toks))
    q_off, q_tok = np.array(offs, np.uint32), np.array(toks, np.uint32)
    ix = _index(fmx, gpu_device, c_off, c_tok, 302, max_tokens=400)
    o = oracle_lib.OracleIndex(max_tokens=400)
    o.add_ids_bulk(c_off, c_tok, 302)
    o.sort(2)
    for kw in [dict(fuzzy=0.6, nmatch=5), dict(fuzzy=0.3, nmatch=3, min_subseq_ratio=0.1)]:
        _check_in_slices(fmx, ix, o, q_off, q_tok, kw, kw["nmatch"], 100, str(kw))


def test_small_path_behind_match_one(fmx, gpu_device):
    """fmx_match_one from eight threads: the coalescer's micro-batches take the one-launch path."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 600, 25)
    ix = _index(fmx, gpu_device, c_off, c_tok, vs)
    o = workloads.build_oracle(c_off, c_tok, vs)
    kw = dict(fuzzy=0.6, nmatch=5)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus())
    errors = []

    def singles(k):
        try:
            prm = fmx.Params(**kw)
            for q in range(k, len(q_off) - 1, 8):
                rec, total = ix.match_one(prm, q_tok[q_off[q]:q_off[q + 1]], cap=5)
                assert len(rec) == oc[q] and rec.tobytes() == oo[q, :oc[q]].tobytes(), (k, q)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=singles, args=(k,)) for k in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:2]
