"""API edge cases through the C ABI on the GPU: one-sentence corpus, empty and single-token patterns, output
capacity below nmatch (truncation + totals), empty batches, duplicate sentences — against the oracle."""
import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu


def _csr(rows):
    off = np.zeros(len(rows) + 1, np.uint32)
    off[1:] = np.cumsum([len(r) for r in rows])
    return off, np.array([t for r in rows for t in r], np.uint32)


def _run(fmx, dev, sents, queries, cap=None, **kw):
    c_off, c_tok = _csr(sents)
    q_off, q_tok = _csr(queries)
    vs = int(max([2] + [t for s in sents for t in s])) + 1
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(dev)
    ix.finalize()
    ix.to_device(dev)
    cap = cap or kw.get("nmatch", 5) or 8
    go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, c_tok, vs)
    o.sort(1)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, max(cap, 64), nthreads=1)
    return ix, go, gc, gt, oo, oc


def test_single_sentence_corpus(fmx, gpu_device):
    _, go, gc, _, oo, oc = _run(fmx, gpu_device, [[2, 3, 4, 5]], [[2, 3, 4, 5], [2, 3, 9, 5], [], [7]],
                                fuzzy=0.5, nmatch=5, min_subseq_length=2)
    assert gc.tolist() == oc.tolist() == [1, 1, 0, 0]
    assert go[0, 0]["score"] == 1.0 and go[1, 0]["score"] == np.float32(0.75)


def test_capacity_below_nmatch_reports_totals(fmx, gpu_device):
    sents = [[2, 3, 4, 5, 6], [2, 3, 4, 5, 7], [2, 3, 4, 5, 8], [2, 3, 4, 9, 6], [2, 3, 4, 5, 6, 7]]
    _, go, gc, gt, oo, oc = _run(fmx, gpu_device, sents, [[2, 3, 4, 5, 6]], cap=2, fuzzy=0.3, nmatch=5, min_subseq_length=2)
    assert gc[0] == 2 and gt[0] == oc[0] == 5
    assert go[0, :2].tobytes() == oo[0, :2].tobytes()


def test_empty_batches_and_patterns(fmx, gpu_device):
    ix, *_ = _run(fmx, gpu_device, [[2, 3, 4]], [[2, 3, 4]], fuzzy=0.5)
    go, gc, gt = ix.match_batch(fmx.Params(), np.zeros(1, np.uint32), np.zeros(0, np.uint32), cap=5)
    assert go.shape[0] == 0 and gc.shape[0] == 0
    go, gc, gt = ix.match_batch(fmx.Params(), np.zeros(4, np.uint32), np.zeros(0, np.uint32), cap=5)
    assert gc.tolist() == [0, 0, 0]


def test_duplicate_sentences_and_one_token_patterns(fmx, gpu_device):
    _, go, gc, _, oo, oc = _run(fmx, gpu_device, [[2, 3, 4]] * 50 + [[5]] * 3, [[2, 3, 4], [5], [3]], cap=64,
                                fuzzy=0.0, nmatch=0, min_subseq_length=1, min_subseq_ratio=0)
    assert gc.tolist() == oc.tolist() == [50, 3, 50]
    assert all(go[i, :gc[i]].tobytes() == oo[i, :oc[i]].tobytes() for i in range(3))


def test_index_adopted_from_arrays_answers_like_the_index_it_came_from(fmx, gpu_device):
    """fmx_index_from_arrays -> finalize -> device: the same records as the index whose arrays were adopted, and as the
    oracle on those arrays."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 500, 20)
    a = fmx.Index.create()
    a.add_ids(c_off, c_tok, vs)
    a.sort(gpu_device)
    a.finalize()
    b = fmx.Index.from_arrays(a.array("suf_sid"), a.array("suf_pos"), a.array("buffer"), a.array("sent_pos"), qva=a.array("qva"))
    b.finalize()
    b.to_device(gpu_device)
    o = oracle_lib.OracleIndex()
    o.set_arrays(b.array("suf_sid"), b.array("suf_pos"), b.array("buffer"), b.array("sent_pos"), b.array("qva"), b.array("sfreq"))
    kw = dict(fuzzy=0.6, nmatch=5)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus())
    go, gc, _ = b.match_batch(fmx.Params(**kw), q_off, q_tok, cap=5)
    workloads.compare(oo, oc, go, gc, "from_arrays")
    assert int((gc > 0).sum()) > 100
