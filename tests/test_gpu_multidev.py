"""The in-library multi-device split (fmx_index_to_devices / fmx_set_devices, csrc/fmx_capi.cc:match_on_devices): a batch
is cut into `replicas x host_lanes` contiguous query blocks, each run by its own host thread on its replica, the records
written straight into the caller's buffers at `out + lo * cap`, penalty-token slots at `q_itok + lo`.  A one-GPU box
cannot hold two ordinals, so the test option "duplicate_devices" lets ONE ordinal carry several replicas: the block
arithmetic, the per-replica contexts and the threading are exactly what an 8-GPU node runs (the reference's N workers on
one const index, cli/src/FuzzyMatch-cli.cc:139-147).  Every record is compared with the oracle."""
import threading

import numpy as np
import pytest

import oracle_lib
import workloads
from test_gpu_realtok import make_sentence, edit, rand_itoks

pytestmark = pytest.mark.gpu


def build_replicated(fmx, dev, c_off, c_tok, vs, replicas):
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(dev)
    ix.finalize()
    ix.set_option("duplicate_devices", 1)
    ix.to_devices([dev] * replicas)
    assert ix.info().on_device == replicas
    return ix


@pytest.fixture(scope="module")
def c2(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(1_000_000, 100_000, 20)
    ref = fmx.Index.create()
    ref.add_ids(c_off, c_tok, vs)
    ref.sort(gpu_device)
    ref.finalize()
    o = oracle_lib.OracleIndex()
    o.set_arrays(ref.array("suf_sid"), ref.array("suf_pos"), ref.array("buffer"), ref.array("sent_pos"), ref.array("qva"), ref.array("sfreq"))
    prm = dict(fuzzy=0.7, nmatch=5)
    oo, oc = o.match_batch(oracle_lib.Params(**prm), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus())
    o.close()
    return c_off, c_tok, q_off, q_tok, vs, prm, oo, oc


@pytest.mark.parametrize("replicas,lanes", [(2, 1), (3, 1), (2, 2), (1, 2)])
def test_c2_queries_over_replicas_and_lanes_match_the_oracle(fmx, gpu_device, c2, replicas, lanes):
    """C2's 100 k queries (BASELINE configs[1]) cut into replicas x lanes blocks: 2, 3, 4 and 2 blocks."""
    c_off, c_tok, q_off, q_tok, vs, prm, oo, oc = c2
    ix = build_replicated(fmx, gpu_device, c_off, c_tok, vs, replicas)
    ix.set_option("host_lanes", lanes)
    ix.set_option("min_block_queries", 4096)
    go, gc, gt = ix.match_batch(fmx.Params(**prm), q_off, q_tok, cap=5)
    workloads.compare(oo, oc, go, gc, f"replicas={replicas} lanes={lanes}")
    # an odd block size, and a batch too small to shard (runs whole on one replica, round-robin)
    n = 33333
    go, gc, gt = ix.match_batch(fmx.Params(**prm), q_off[:n + 1], q_tok[:int(q_off[n])], cap=5)
    workloads.compare(oo[:n], oc[:n], go, gc, "odd blocks")
    for _ in range(replicas + 1):
        go, gc, gt = ix.match_batch(fmx.Params(**prm), q_off[:501], q_tok[:int(q_off[500])], cap=5)
        workloads.compare(oo[:500], oc[:500], go, gc, "small batch on one replica")


def test_host_lanes_set_before_the_upload(fmx, gpu_device, c2):
    """host_lanes set before fmx_index_to_devices reaches the replicas created later (their context pools)."""
    c_off, c_tok, q_off, q_tok, vs, prm, oo, oc = c2
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.set_option("host_lanes", 3)
    ix.set_option("min_block_queries", 1000)
    ix.set_option("duplicate_devices", 1)
    ix.to_devices([gpu_device, gpu_device])
    go, gc, gt = ix.match_batch(fmx.Params(**prm), q_off, q_tok, cap=5)   # 6 blocks
    workloads.compare(oo, oc, go, gc, "2 replicas x 3 lanes")


def test_match_one_over_replicas(fmx, gpu_device, c2):
    """fmx_match_one from many threads over three replicas: the coalescer's micro-batches run whole on one replica each."""
    c_off, c_tok, q_off, q_tok, vs, prm, oo, oc = c2
    ix = build_replicated(fmx, gpu_device, c_off, c_tok, vs, 3)
    ix.set_option("small_path", 1)
    params = fmx.Params(**prm)
    n_threads, per = 16, 40
    errs = []

    def worker(t):
        try:
            for k in range(per):
                q = (t * 9973 + k * 131) % (len(q_off) - 1)
                out, tot = ix.match_one(params, q_tok[q_off[q]:q_off[q + 1]], cap=5)
                assert len(out) == oc[q] and out.tobytes() == oo[q, :oc[q]].tobytes(), q
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]


def test_real_tokens_and_penalty_tokens_over_blocks(fmx, gpu_device):
    """q_real / q_itok present: the penalty-token slot of (query q, position j) is q_off[q] + q + j, so a block's view of
    q_itok starts `lo` slots in (fmx_capi.cc: `s.q_itok = v.q_itok + lo`).  2 replicas x 2 lanes over 400 queries."""
    rng = np.random.default_rng(11)
    ix = fmx.Index.create(pt=49)
    o = oracle_lib.OracleIndex()
    sents = [make_sentence(rng) for _ in range(300)]
    for i, (norm, real, itoks) in enumerate(sents):
        assert ix.add_tm(str(i), norm, real=real, itoks=itoks) == i
        o.add_tm(norm, real=real, itoks=itoks)
    ix.sort(gpu_device)
    ix.finalize()
    ix.set_option("duplicate_devices", 1)
    ix.set_option("host_lanes", 2)
    ix.set_option("min_block_queries", 16)
    ix.to_devices([gpu_device, gpu_device])
    o.sort(2)
    queries = []
    for _ in range(400):
        norm, real, _ = sents[int(rng.integers(0, len(sents)))]
        norm, real = edit(rng, norm, real)
        queries.append((norm, real, rand_itoks(rng, len(norm))))
    q_off = np.zeros(len(queries) + 1, np.uint32)
    q_off[1:] = np.cumsum([len(q[0]) for q in queries])
    q_tok = np.concatenate([ix.vocab_lookup(q[0]) for q in queries]).astype(np.uint32)
    q_real = np.asarray([ix.lookup_real(r) for q in queries for r in q[1]], np.uint32)
    q_itok = np.zeros(int(q_off[-1]) + len(queries), np.uint32)
    for qi, (norm, real, itoks) in enumerate(queries):
        for k, v in itoks.items():
            q_itok[int(q_off[qi]) + qi + k] = ix.intern_itok(v)
    n_checked = 0
    for kw in [dict(fuzzy=0.5, nmatch=8, min_subseq_length=2, min_subseq_ratio=0),
               dict(fuzzy=0.4, nmatch=4, min_subseq_length=2, min_subseq_ratio=0, contrast=0.5, contrast_buffer=6)]:
        go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, q_real=q_real, q_itok=q_itok, cap=64)
        oprm = oracle_lib.Params(**kw)
        for qi, (norm, real, itoks) in enumerate(queries):
            ref, total = o.match_ids(oprm, q_tok[q_off[qi]:q_off[qi + 1]], real=real, itoks=itoks, cap=64)
            got = go[qi, :gc[qi]]
            assert len(ref) == len(got) and ref.tobytes() == got.tobytes(), (kw, qi, norm, real, itoks, ref, got)
            n_checked += len(ref)
    assert n_checked > 800, n_checked


def test_concurrent_interning_and_matching(fmx, gpu_device):
    """Penalty strings interned by one thread while others match (the facade interns per call, fmx_index_intern_itok):
    the distance table is re-laid out under the same mutex a replica snapshots it under."""
    rng = np.random.default_rng(12)
    ix = fmx.Index.create(pt=49)
    sents = [make_sentence(rng) for _ in range(200)]
    for i, (norm, real, itoks) in enumerate(sents):
        ix.add_tm(str(i), norm, real=real, itoks=itoks)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    norm = sents[0][0]
    wids = ix.vocab_lookup(norm).astype(np.uint32)
    q_real = np.asarray([ix.lookup_real(r) for r in sents[0][1]], np.uint32)
    prm = fmx.Params(fuzzy=0.3, nmatch=4, min_subseq_length=1, min_subseq_ratio=0)
    base, _ = ix.match_one(prm, wids, q_real=q_real, cap=8)
    stop = threading.Event()
    errs = []

    def interner():
        k = 0
        while not stop.is_set() and k < 600:
            ix.intern_itok(f"s{k}")
            k += 1

    def matcher():
        try:
            for _ in range(150):
                out, _ = ix.match_one(prm, wids, q_real=q_real, cap=8)
                assert out.tobytes() == base.tobytes()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=interner)] + [threading.Thread(target=matcher) for _ in range(4)]
    for t in th:
        t.start()
    for t in th[1:]:
        t.join()
    stop.set()
    th[0].join()
    assert not errs, errs[:3]
