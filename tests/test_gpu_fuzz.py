"""Randomised parameter sweep: GPU (through the C ABI) against the oracle on small synthetic corpora, bit for
bit.  Every case draws its own parameters (threshold, nmatch, no_perfect, subsequence limits, idf, edit costs,
contrastive rerank with both reductions and several buffers), so the rarely combined branches meet."""
import os

import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("FMX_FUZZ_CASES", "48"))


def _draw(rng):
    kw = dict(fuzzy=float(rng.choice([0.0, 0.3, 0.5, 0.6, 0.7, 0.8, 0.95])), nmatch=int(rng.choice([0, 1, 2, 3, 5, 8, 17])),
              no_perfect=bool(rng.random() < 0.3), min_subseq_length=int(rng.choice([1, 2, 3, 4])),
              min_subseq_ratio=float(rng.choice([0.0, 0.1, 0.3, 0.5])))
    r = rng.random()
    if r < 0.25:
        kw["idf_penalty"] = float(rng.choice([0.3, 1.0]))
    elif r < 0.5:
        # replace > 0 and insert + delete > 0 keep Costs::get_normalizer positive; with a zero normaliser the
        # reference computes on NaN and converts it to int (undefined), which is outside the parity contract
        ins, dele = (float(x) for x in rng.choice([0.0, 0.5, 1.0, 2.0], size=2))
        if ins + dele == 0.0:
            ins = 1.0
        kw["costs"] = (ins, dele, float(rng.choice([0.5, 1.0, 2.0])))
    if rng.random() < 0.35:
        kw["contrast"] = float(rng.choice([0.1, 0.3, 0.8]))
        kw["reduce"] = int(rng.integers(0, 2))
        kw["contrast_buffer"] = int(rng.choice([-1, 0, 3, 32]))
    return kw


@pytest.fixture(scope="module")
def corpora(fmx, gpu_device):
    out = []
    for seed, (n_sent, mean_len) in enumerate([(6000, 12), (4000, 25)]):
        c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(n_sent, 600, mean_len)
        ix = fmx.Index.create()
        ix.add_ids(c_off, c_tok, vs)
        ix.sort(gpu_device)
        ix.finalize()
        ix.to_device(gpu_device)
        o = workloads.build_oracle(c_off, c_tok, vs)
        out.append((ix, o, q_off, q_tok))
    return out


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_parameters(fmx, corpora, case):
    rng = np.random.default_rng(1000 + case)
    ix, o, q_off, q_tok = corpora[case % len(corpora)]
    kw = _draw(rng)
    cap = kw["nmatch"] or 128
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
    go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
    workloads.compare(oo, oc, go, gc, str(kw))


@pytest.fixture(scope="module")
def dense(fmx, gpu_device):
    """Tiny vocabularies: hundreds of candidates per query, many exact ties in score and match length, UNK /
    empty / over-long / id-0 patterns (workloads.edge_queries)."""
    out = []
    for seed, vocab in enumerate([3, 5, 9]):
        rng = np.random.default_rng(50 + seed)
        c_off, c_tok = workloads.dense_corpus(rng, 1500, vocab, min_len=1, max_len=14)
        q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, vocab, 300)
        ix = fmx.Index.create(max_tokens=24)
        ix.add_ids(c_off, c_tok, vocab + 2)
        ix.sort(gpu_device)
        ix.finalize()
        ix.to_device(gpu_device)
        o = oracle_lib.OracleIndex(max_tokens=24)
        o.add_ids_bulk(c_off, c_tok, vocab + 2)
        o.sort(2)
        out.append((ix, o, q_off, q_tok))
    return out


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_parameters_dense(fmx, dense, case):
    rng = np.random.default_rng(5000 + case)
    ix, o, q_off, q_tok = dense[case % len(dense)]
    kw = _draw(rng)
    cap = kw["nmatch"] or 2048
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
    go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
    workloads.compare(oo, oc, go, gc, str(kw))
