"""The expand stage's layouts and the host-buffer path's scheduling must not change a single record: light queries on /
off / at the largest setting, both kernel orders, a serialised pass, one or two queries per wave in the last kernel, and large concurrent host-buffer calls that take
turns on the GPU (src/fuzzy_match.cc:484-612 is order-free up to the candidate list; the reference's match() is const
and re-entrant, cli/src/FuzzyMatch-cli.cc:139-147)."""
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
    return ix


@pytest.mark.parametrize("corpus", ["synthetic", "dense"])
def test_expand_layouts_agree_with_the_oracle(fmx, gpu_device, corpus):
    if corpus == "synthetic":
        c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(30000, 1500, 25)
        ix = _index(fmx, gpu_device, c_off, c_tok, vs)
        o = workloads.build_oracle(c_off, c_tok, vs)
        sets = ["C3", "C4", "wide_all", "contrast_mean_all"]
    else:
        rng = np.random.default_rng(77)
        c_off, c_tok = workloads.dense_corpus(rng, 3000, 6, min_len=1, max_len=14)
        q_off, q_tok = workloads.edge_queries(rng, c_off, c_tok, 6, 400)
        ix = _index(fmx, gpu_device, c_off, c_tok, 8, max_tokens=24)
        o = oracle_lib.OracleIndex(max_tokens=24)
        o.add_ids_bulk(c_off, c_tok, 8)
        o.sort(2)
        sets = ["wide_all", "fuzzy_zero", "costs_101", "ml_negative"]
    # light_flat: the light queries of at most that many hits several to a wave (k_expand_lf, round 5: measured, slower, off by
    # default — profiles/r05_light_flat_ab.md — and kept bit-identical here: every light query, only the small ones, with the rest
    # of the light queries beside it in the round-4 kernel)
    layouts = [dict(light_max=256, expand_order=1), dict(light_max=0, expand_order=1), dict(light_max=1024, expand_order=0),
               dict(light_max=17, expand_order=0), dict(light_max=256, expand_order=1, serial=1), dict(light_max=256, expand_order=0, heavy_min=0),
               dict(light_max=256, expand_order=1, pair_finish=0),  # the others run k_finish_pair where the parameters allow it
               dict(light_max=256, expand_order=1, light_flat=256), dict(light_max=256, expand_order=0, light_flat=128),
               dict(light_max=256, expand_order=1, light_flat=9),
               # the probe as ONE kernel (round 4's layout; the default since round 5 is head + tail, k_probe + k_probe_tail)
               dict(light_max=256, expand_order=1, probe_split=0), dict(light_max=0, expand_order=0, probe_split=0)]
    light_seen = 0
    for name in sets:
        kw = workloads.PARAM_SETS[name]
        cap = kw["nmatch"] or 256
        oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, cap, nthreads=oracle_lib.effective_cpus())
        for lay in layouts:
            for k, v in {"serial": 0, "heavy_min": 1024, "pair_finish": 1, "light_flat": 0, "probe_split": 1, **lay}.items():
                ix.set_option(k, v)
            go, gc, gt, st = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap, want_stats=True)
            workloads.compare(oo, oc, go, gc, f"{corpus} {name} {lay}")
            if lay["light_max"] == 0:
                assert st.n_light_hits == 0
            light_seen += int(st.n_light_hits)
    assert light_seen > 0  # the light path did run
    ix.set_option("light_flat", 0)
    ix.set_option("probe_split", 1)


def test_flat_light_walk_is_bit_identical(fmx, gpu_device):
    """k_expand_lf on a batch large enough for tiles of 8, 16 and 32 queries (the grid's choice by pass size) and for queries that
    keep more than one wave of hits (left unreduced in their segment for k_cand_route): every record equals the oracle's."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(120_000, 140_000, 22)
    ix = _index(fmx, gpu_device, c_off, c_tok, vs)
    o = workloads.build_oracle(c_off, c_tok, vs)
    for name, nq in (("C3", 140_000), ("wide_all", 20_000), ("C4", 9_000)):
        kw = workloads.PARAM_SETS[name]
        cap = kw["nmatch"] or 256
        qo, qt = q_off[:nq + 1], q_tok[:int(q_off[nq])]
        oo, oc = o.match_batch(oracle_lib.Params(**kw), qo, qt, cap, nthreads=oracle_lib.effective_cpus())
        for flat in (256, 128, 40):
            ix.set_option("light_flat", flat)
            go, gc, gt, st = ix.match_batch(fmx.Params(**kw), qo, qt, cap=cap, want_stats=True)
            workloads.compare(oo, oc, go, gc, f"{name} light_flat={flat}")
            assert st.n_light_hits > 0
    ix.set_option("light_flat", 0)
    o.close()


def test_large_concurrent_calls_take_turns(fmx, gpu_device):
    """Three threads, each with calls of 40 000 queries (>= the 32 768 from which host-buffer calls take turns on the
    kernels), from pinned and from pageable buffers; every record equals the oracle's."""
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20000, 40000, 20)
    ix = _index(fmx, gpu_device, c_off, c_tok, vs)
    ix.set_option("contexts", 3)
    o = workloads.build_oracle(c_off, c_tok, vs)
    kw = dict(fuzzy=0.6, nmatch=3)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 3, nthreads=oracle_lib.effective_cpus())
    po, pt = fmx.pinned_copy(q_off.astype(np.uint32)), fmx.pinned_copy(q_tok.astype(np.uint32))
    errors = []

    def worker(k):
        try:
            out = (fmx.pinned_empty((len(oc), 3), fmx.MATCH_DTYPE), fmx.pinned_empty(len(oc), np.uint32), fmx.pinned_empty(len(oc), np.uint32)) if k != 1 else None
            for _ in range(4):
                go, gc, gt = ix.match_batch(fmx.Params(**kw), po if k != 1 else q_off, pt if k != 1 else q_tok, cap=3, out=out)
                workloads.compare(oo, oc, go, gc, f"thread {k}")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:2]
    ix.set_option("turns", 0)  # and without turns
    go, gc, gt = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=3)
    workloads.compare(oo, oc, go, gc, "no turns")
