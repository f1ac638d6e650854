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
    o = workloads.build_oracle(c_off, c_tok, vs, sfreq=ix.array("sfreq"), nthreads=4)
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
    o.add_ids_bulk(c_off, c_tok, vocab + 2, sfreq=ix.array("sfreq"))
    o.sort(2)
    em = emul_lib.Emul(ix)
    for name, kw in workloads.PARAM_SETS.items():
        cap = kw.get("nmatch", 5) or 512
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=2)
        eo, ec, _ = em.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        workloads.compare(oo, oc, eo, ec, f"{name}/seed{seed}")
