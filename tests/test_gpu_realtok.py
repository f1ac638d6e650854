"""Real-token / penalty-token cost model on the GPU (k_dp<FULL>: src/edit_distance.cc:19-38,53-62,
include/fuzzy/edit_distance.hxx:7-35) against the oracle, BIT FOR BIT, on randomised string-level corpora:
case features (L U M C first bytes), digit real tokens behind ｟num｠, real tokens equal to / different from the
normalised word, empty real tokens (also trailing: the split drops it, src/sentence.cc:19-20), penalty strings of
one and several bytes at any slot including the trailing one.  Also: batches of changing shape on one index (the
staging buffers grow independently), patterns longer than the LDS rows hold (rows in HBM), concurrent callers."""
import threading

import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu

WORDS = ["aa", "bb", "cc", "dd", "ee", "ff", "gg", "hh", "ii", "jj", "kk", "｟num｠", ".", "ｘｙ"]
ITOKS = ["T", " ", "｟it｠", "Tx", "é", "TT", ""]


def real_for(rng, word):
    r = rng.random()
    if word == "｟num｠":
        return str(int(rng.integers(0, 200)))           # the digits (pt_nbr)
    if r < 0.45:
        return "L"                                       # case feature letters: cost rep * 1
    if r < 0.6:
        return ["U", "C", "M", "N"][int(rng.integers(0, 4))]
    if r < 0.8:
        return word                                      # identical to the normalised form
    if r < 0.9:
        return word.upper() if word.isascii() else "Ｘ"  # a different surface form: cost rep * 2
    return ""                                            # empty real token


def rand_itoks(rng, n):
    out = {}
    for _ in range(int(rng.integers(0, 3))):
        s = ITOKS[int(rng.integers(0, len(ITOKS)))]
        if s:
            out[int(rng.integers(0, n + 1))] = s         # slot n = the trailing penalty string
    return out


def make_sentence(rng, lo=1, hi=9):
    n = int(rng.integers(lo, hi + 1))
    norm = [WORDS[int(rng.integers(0, len(WORDS)))] for _ in range(n)]
    real = [real_for(rng, w) for w in norm]
    if not real[0]:
        real[0] = "L"  # a sentence whose real-token string is empty is ignored by add_tm (src/suffix_array_index.cc:16)
    return norm, real, rand_itoks(rng, n)


def edit(rng, norm, real):
    norm, real = list(norm), list(real)
    for _ in range(int(rng.integers(0, 3))):
        k = int(rng.integers(0, 3))
        i = int(rng.integers(0, len(norm)))
        if k == 0 and len(norm) > 1:
            del norm[i], real[i]
        elif k == 1:
            w = WORDS[int(rng.integers(0, len(WORDS)))] if rng.random() < 0.8 else "zz"  # zz: out of vocabulary
            norm[i], real[i] = w, real_for(rng, w)
        else:
            w = WORDS[int(rng.integers(0, len(WORDS)))]
            norm.insert(i, w)
            real.insert(i, real_for(rng, w))
    for i in range(len(real)):                            # same words, other real tokens: the real-token term alone
        if rng.random() < 0.3:
            real[i] = real_for(rng, norm[i])
    return norm, real


PARAMS = [dict(fuzzy=0.5, nmatch=8, min_subseq_length=2, min_subseq_ratio=0),
          dict(fuzzy=0.3, nmatch=0, min_subseq_length=1, min_subseq_ratio=0, no_perfect=True),
          dict(fuzzy=0.4, nmatch=5, min_subseq_length=2, min_subseq_ratio=0, idf_penalty=0.6),
          dict(fuzzy=0.3, nmatch=6, min_subseq_length=2, min_subseq_ratio=0, costs=(0.5, 1, 2)),
          dict(fuzzy=0.4, nmatch=4, min_subseq_length=2, min_subseq_ratio=0, contrast=0.5, contrast_buffer=6)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_real_tokens_and_itoks_bit_exact(fmx, gpu_device, seed):
    rng = np.random.default_rng(seed)
    ix = fmx.Index.create(pt=49)
    o = oracle_lib.OracleIndex()
    sents = [make_sentence(rng) for _ in range(300)]
    for i, (norm, real, itoks) in enumerate(sents):
        assert ix.add_tm(str(i), norm, real=real, itoks=itoks) == i
        o.add_tm(norm, real=real, itoks=itoks)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o.sort(2)
    assert ix.info().real_differs
    queries = []
    for _ in range(250):
        norm, real, _ = sents[int(rng.integers(0, len(sents)))]
        norm, real = edit(rng, norm, real)
        if rng.random() < 0.15 and real:
            real[-1] = ""                                 # trailing empty real token
        queries.append((norm, real, rand_itoks(rng, len(norm))))
    q_off = np.zeros(len(queries) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(q[0]) for q in queries])
    q_tok = np.concatenate([ix.vocab_lookup(q[0]) for q in queries]).astype(np.uint32)
    q_real = np.asarray([ix.lookup_real(r) for q in queries for r in q[1]], np.uint32)
    q_itok = np.zeros(int(q_off[-1]) + len(queries), np.uint32)
    for qi, (norm, real, itoks) in enumerate(queries):
        for k, v in itoks.items():
            q_itok[int(q_off[qi]) + qi + k] = ix.intern_itok(v)
    n_checked = n_real_cost = 0
    for kw in PARAMS:
        cap = 64
        go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, q_real=q_real, q_itok=q_itok, cap=cap)
        oprm = oracle_lib.Params(**kw)
        for qi, (norm, real, itoks) in enumerate(queries):
            ref, total = o.match_ids(oprm, q_tok[q_off[qi]:q_off[qi + 1]], real=real, itoks=itoks, cap=cap)
            got = go[qi, :gc[qi]]
            assert len(ref) == len(got) and ref.tobytes() == got.tobytes(), (seed, kw, qi, norm, real, itoks, ref, got)
            n_checked += len(ref)
            # scores that are neither 1 nor a multiple of 1/len: the real-token / penalty-token terms were active
            n_real_cost += int(np.sum((ref["score"] < 1) & (ref["score"] > 0.9)))
    assert n_checked > 2000 and n_real_cost > 50, (n_checked, n_real_cost)


def test_changing_batch_shapes_on_one_index(fmx, gpu_device):
    """Many short queries, then a few long ones, then many again: every staging buffer keeps its own capacity."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 600, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, vs)
    kw = dict(fuzzy=0.6, nmatch=5)
    lens = np.diff(q_off)
    short = np.argsort(lens)[:400]
    long_ = np.argsort(lens)[-12:]

    def sub(idx):
        off = np.zeros(len(idx) + 1, np.uint32)
        off[1:] = np.cumsum(lens[idx])
        tok = np.concatenate([q_tok[q_off[i]:q_off[i + 1]] for i in idx]).astype(np.uint32)
        return off, tok

    for name, idx in [("short", short), ("long", long_), ("short_again", short[:50]), ("all", np.arange(len(lens)))]:
        off, tok = sub(idx)
        oo, oc = o.match_batch(oracle_lib.Params(**kw), off, tok, 5, nthreads=oracle_lib.effective_cpus())
        go, gc, gt = ix.match_batch(fmx.Params(**kw), off, tok, cap=5)
        workloads.compare(oo, oc, go, gc, name)


def test_patterns_beyond_the_lds_rows(fmx, gpu_device):
    """--max-tokens-in-pattern above the default (include/fuzzy/suffix_array_index.hh:15): patterns of ~700 and
    ~1500 tokens run the same DP over rows in HBM, and the contrastive rerank of such sentences too."""
    rng = np.random.default_rng(5)
    vocab = 30
    lens = np.concatenate([rng.integers(650, 800, 12), rng.integers(1400, 1600, 4), rng.integers(5, 40, 100)])
    c_off = np.zeros(len(lens) + 1, np.uint32)
    c_off[1:] = np.cumsum(lens)
    c_tok = rng.integers(2, vocab + 2, int(c_off[-1])).astype(np.uint32)
    qs = []
    for s in list(range(16)) + list(rng.integers(16, len(lens), 30)):
        t = c_tok[c_off[s]:c_off[s + 1]].copy()
        n_edit = int(rng.integers(0, max(1, len(t) // 12) + 1))
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
    # min_subseq_ratio far below the default: an edit every ~12 tokens leaves no common run of 0.3 * 700 tokens
    mr = dict(min_subseq_length=3, min_subseq_ratio=0.01)
    for name, kw in [("f05", dict(fuzzy=0.5, nmatch=5, **mr)), ("idf", dict(fuzzy=0.4, nmatch=4, idf_penalty=0.5, **mr)),
                     ("costs", dict(fuzzy=0.3, nmatch=5, costs=(0.2, 0.2, 0.7), **mr)),
                     ("contrast", dict(fuzzy=0.3, nmatch=3, contrast=0.4, contrast_buffer=8, **mr)),
                     ("contrast_all", dict(fuzzy=0.3, nmatch=0, contrast=0.4, contrast_buffer=0, **mr)),
                     ("default_mr", dict(fuzzy=0.5, nmatch=5))]:
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 16, nthreads=oracle_lib.effective_cpus())
        go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=16)
        workloads.compare(oo, oc, go, gc, name)
        if name not in ("default_mr", "idf"):
            assert int(oc[:16].min()) >= 1, name  # every long query finds at least its own sentence


def test_concurrent_callers_and_coalescer(fmx, gpu_device):
    """match() is const and re-entrant in the reference (cli/src/FuzzyMatch-cli.cc:139-147): batches from several
    threads at once, and one-pattern calls through the coalescer, give the oracle's records."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 800, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, vs)
    sets = [dict(fuzzy=0.6, nmatch=5), dict(fuzzy=0.5, nmatch=3, idf_penalty=0.5)]
    refs = [o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus()) for kw in sets]
    errors = []

    def batches(k):
        try:
            for _ in range(4):
                go, gc, gt = ix.match_batch(fmx.Params(**sets[k % 2]), q_off, q_tok, cap=5)
                workloads.compare(refs[k % 2][0], refs[k % 2][1], go, gc, f"thread{k}")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def singles(k):
        try:
            prm = fmx.Params(**sets[k % 2])
            oo, oc = refs[k % 2]
            for q in range(k, len(q_off) - 1, 8):
                rec, total = ix.match_one(prm, q_tok[q_off[q]:q_off[q + 1]], cap=5)
                assert len(rec) == oc[q] and rec.tobytes() == oo[q, :oc[q]].tobytes(), (k, q)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=batches, args=(k,)) for k in range(4)] + [threading.Thread(target=singles, args=(k,)) for k in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:2]
