"""fmx_match_batch_device: every buffer already in HBM.  With statistics every
pass is synchronised and timed; without, the passes run back to back and concurrent callers — one host thread and stream
each — are chained on the GPU.  Both against the oracle, bit for bit."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib
import workloads

pytestmark = pytest.mark.gpu


class Hip:
    """Device buffers and streams straight from the HIP runtime libfmx.so is linked against (no torch: a second HIP runtime
    initialised after the first one finds no device)."""

    def __init__(self):
        self.L = C.CDLL("/opt/rocm/lib/libamdhip64.so")

    def check(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self.check(self.L.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 4))))
        self.check(self.L.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1))
        return p.value

    def zeros(self, nbytes):
        p = C.c_void_p()
        self.check(self.L.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 4))))
        self.check(self.L.hipMemset(p, 0, C.c_size_t(nbytes)))
        # hipMemset returns before the fill has run, and the non-blocking streams the calls use are not ordered behind the null
        # stream: with every hardware queue busy the fill could land AFTER a call's kernels had written their records (seen once
        # in 144 calls: a whole call's counts zero).  Wait for the null stream.
        self.check(self.L.hipStreamSynchronize(None))
        return p.value

    def download(self, ptr, dtype, count):
        out = np.zeros(count, dtype=dtype)
        self.check(self.L.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2))
        return out

    def stream(self):
        s = C.c_void_p()
        self.check(self.L.hipStreamCreateWithFlags(C.byref(s), 1))  # hipStreamNonBlocking
        return s.value

    def sync(self):
        self.check(self.L.hipDeviceSynchronize())


@pytest.fixture(scope="module")
def setup(fmx, gpu_device):
    hip = Hip()
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(200_000, 60_000, 20)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    ix.set_option("contexts", 4)
    o = oracle_lib.OracleIndex()
    o.set_arrays(ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos"), ix.array("qva"), int(ix.info().vocab_size))   # (the oracle counts the sentence frequencies itself)
    kw = dict(fuzzy=0.7, nmatch=5)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, q_tok, 5, nthreads=oracle_lib.effective_cpus())
    o.close()
    d_off = hip.upload(q_off.astype(np.uint32))
    d_tok = hip.upload(q_tok.astype(np.uint32))
    return hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc


def _records(hip, d_out, d_cnt, nq, cap):
    return hip.download(d_out, oracle_lib.MATCH_DTYPE, nq * cap).reshape(nq, cap), hip.download(d_cnt, np.uint32, nq)


@pytest.mark.parametrize("want_stats", [True, False])
@pytest.mark.parametrize("sub_queries", [0, 7001])
def test_device_entry_point_matches_the_oracle(fmx, setup, want_stats, sub_queries):
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    nq, cap = len(q_off) - 1, 5
    if sub_queries:
        ix.set_option("sub_queries", sub_queries)   # several passes per call, bounds read from the device
    try:
        d_out, d_cnt, st = hip.zeros(nq * cap * 16), hip.zeros(nq * 4), hip.stream()
        hip.sync()
        ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), d_out, d_cnt, 0,
                              cap, stream=st, max_len=int(np.diff(q_off).max()), want_stats=want_stats)
        go, gc = _records(hip, d_out, d_cnt, nq, cap)
        workloads.compare(oo, oc, go, gc, f"device entry point stats={want_stats} sub={sub_queries}")
    finally:
        ix.set_option("sub_queries", 1 << 20)


@pytest.mark.parametrize("turns", [0, 1, 2])
def test_concurrent_device_calls_are_chained(fmx, setup, turns):
    """Four host threads, each on its own stream, three calls each: whatever the chaining mode (none, behind the predecessor's
    last kernel, behind its memory-bound half) every call leaves the oracle's records."""
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    nq, cap = len(q_off) - 1, 5
    ix.set_option("turns", turns)
    errs = []

    def worker(k):
        try:
            st = hip.stream()
            for _ in range(3):
                d_out, d_cnt = hip.zeros(nq * cap * 16), hip.zeros(nq * 4)
                ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), d_out, d_cnt, 0, cap, stream=st,
                                      max_len=int(np.diff(q_off).max()), want_stats=False)
                go, gc = _records(hip, d_out, d_cnt, nq, cap)
                workloads.compare(oo, oc, go, gc, f"thread {k} turns={turns}")
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)[:500])

    try:
        th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        ix.set_option("turns", 1)
    assert not errs, errs[:2]


@pytest.mark.parametrize("turns", [1, 2])
def test_begin_end_keeps_several_calls_in_flight_from_one_thread(fmx, setup, turns):
    """fmx_match_batch_device_begin / _end: three calls pending at once, enqueued by this thread, ended in order; further
    begins without a free context are deferred on the host (round 5) and started as the earlier calls end — not refused, not
    dead-locked."""
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    nq, cap = len(q_off) - 1, 5
    ix.set_option("turns", turns)
    ix.set_option("contexts", 3)
    try:
        bufs = [(hip.zeros(nq * cap * 16), hip.zeros(nq * 4)) for _ in range(3)]
        hip.sync()
        ml = int(np.diff(q_off).max())
        # the pool is finite: eight begins, at most as many running as there are contexts, the rest queued on the host in order
        more = [(hip.zeros(nq * cap * 16), hip.zeros(nq * 4)) for _ in range(8)]
        hip.sync()
        pend = [ix.device_begin(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), o, c, 0, cap, stream=0, max_len=ml) for o, c in more]
        ix.device_end(pend.pop())     # the LAST one first: nothing has freed a slot yet, it starts itself
        for h in pend:
            ix.device_end(h)
        for o, c in more:
            go, gc = _records(hip, o, c, nq, cap)
            workloads.compare(oo, oc, go, gc, f"deferred begin turns={turns}")
        for rnd in range(2):
            pend = [ix.device_begin(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), o, c, 0, cap, stream=0, max_len=ml) for o, c in bufs]
            for h in pend:
                ix.device_end(h)
            for o, c in bufs:
                go, gc = _records(hip, o, c, nq, cap)
                workloads.compare(oo, oc, go, gc, f"begin/end turns={turns} round {rnd}")
    finally:
        ix.set_option("turns", 1)
        ix.set_option("contexts", 4)


def test_device_turns_is_a_separate_option(fmx, setup):
    """`device_turns` chains the device-resident entry points only (default 0: they overlap freely); `turns` keeps setting
    both.  Four concurrent callers under either setting leave the oracle's records."""
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    nq, cap = len(q_off) - 1, 5
    errs = []

    def worker(k):
        try:
            st = hip.stream()
            d_out, d_cnt = hip.zeros(nq * cap * 16), hip.zeros(nq * 4)
            ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), d_out, d_cnt, 0, cap, stream=st,
                                  max_len=int(np.diff(q_off).max()), want_stats=False)
            go, gc = _records(hip, d_out, d_cnt, nq, cap)
            workloads.compare(oo, oc, go, gc, f"thread {k}")
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)[:500])

    try:
        for dt in (2, 0):
            ix.set_option("device_turns", dt)
            th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
            for t in th:
                t.start()
            for t in th:
                t.join()
    finally:
        ix.set_option("turns", 1)
    assert not errs, errs[:2]
    with pytest.raises(Exception):
        ix.set_option("no_such_option", 1)


def test_rerank_distances_are_counted(fmx, setup):
    """fmx_stats.n_rerank: sentence-pair distances of the flattened contrastive rerank — none without a contrastive factor."""
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    n = 5000
    qo, qt = q_off[:n + 1], q_tok[:int(q_off[n])]
    plain = ix.match_batch(fmx.Params(**kw), qo, qt, cap=5, want_stats=True)[3]
    assert plain.n_rerank == 0
    st = ix.match_batch(fmx.Params(contrast=0.3, contrast_buffer=32, **kw), qo, qt, cap=5, want_stats=True)[3]
    assert st.n_rerank > 0


def test_a_wrong_total_tokens_is_refused(fmx, setup):
    """total_tokens is checked on the device against q_off (k_setup): a stale or approximate value fails the call instead of
    running the passes with wrong token bounds."""
    hip, ix, kw, q_off, q_tok, d_off, d_tok, oo, oc = setup
    nq, cap = 20_000, 5
    d_out, d_cnt = hip.zeros(nq * cap * 16), hip.zeros(nq * 4)
    hip.sync()
    ml = int(np.diff(q_off).max())
    # (the last one overstates the batch far beyond the token buffer the caller uploaded: the pass must be inert on the device —
    # k_probe may not walk tokens no query owns — not only refused afterwards; round-4 advice)
    for wrong in (int(q_off[nq]) - 7, int(q_off[nq]) + 5, 3 * int(q_off[-1]) + 1_000_003):
        with pytest.raises(fmx.FmxError):
            ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, wrong, d_out, d_cnt, 0, cap, stream=0, max_len=ml, want_stats=False)
        h = ix.device_begin(fmx.Params(**kw), nq, d_off, d_tok, wrong, d_out, d_cnt, 0, cap, stream=0, max_len=ml)
        with pytest.raises(fmx.FmxError):
            ix.device_end(h)
    # and the right value still works afterwards, in the same contexts
    ix.match_batch_device(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[nq]), d_out, d_cnt, 0, cap, stream=0, max_len=ml, want_stats=False)
    go, gc = _records(hip, d_out, d_cnt, nq, cap)
    workloads.compare(oo[:nq], oc[:nq], go, gc, "after refused calls")


def test_tables_are_refreshed_while_this_thread_has_calls_pending(fmx, gpu_device):
    """The penalty-string table and the spelling table are pinned per call, not locked across it: a thread that holds pending
    calls can intern a new penalty string, make its first text batch and start further calls without dead-locking against
    itself (round-3 advice)."""
    import faulthandler
    hip = Hip()
    faulthandler.dump_traceback_later(120, exit=True)   # a dead-lock ends the test run instead of hanging the box
    try:
        c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(20_000, 4_000, 12)
        ix = fmx.Index.create()
        ix.add_ids(c_off, c_tok, vs)
        ix.sort(gpu_device)
        ix.finalize()
        ix.to_device(gpu_device)
        ix.set_option("contexts", 4)
        words = [b"", b""] + [b"w%d" % k for k in range(2, vs)]
        arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
        offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
        pos = 0
        for k in range(2, len(words)):
            offs[k] = pos
            pos += len(words[k]) + 1
        ix.set_forms_packed(arena, offs)
        kw = dict(fuzzy=0.6, nmatch=3)
        nq, cap = len(q_off) - 1, 3
        d_off, d_tok = hip.upload(q_off.astype(np.uint32)), hip.upload(q_tok.astype(np.uint32))
        bufs = [(hip.zeros(nq * cap * 16), hip.zeros(nq * 4)) for _ in range(2)]
        hip.sync()
        ml = int(np.diff(q_off).max())
        ref = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        pend = [ix.device_begin(fmx.Params(**kw), nq, d_off, d_tok, int(q_off[-1]), o, c, 0, cap, stream=0, max_len=ml) for o, c in bufs]
        # with two calls pending: a new penalty string (the distance table grows), then a call that refreshes the device table
        # and the first text batch of the index (uploads the spelling table)
        assert ix.intern_itok("｟new_tag｠") > 0
        again = ix.match_batch(fmx.Params(**kw), q_off, q_tok, cap=cap)
        text = np.frombuffer(b"".join(words[t] + b"\0" for t in q_tok), dtype=np.uint8).copy()
        got = ix.match_batch_text(fmx.Params(**kw), q_off, text, cap=cap)
        for h in pend:
            ix.device_end(h)
        for o, c in bufs:
            go, gc = _records(hip, o, c, nq, cap)
            workloads.compare(ref[0], ref[1], go, gc, "pending call")
        workloads.compare(ref[0], ref[1], again[0], again[1], "call issued while others were pending")
        workloads.compare(ref[0], ref[1], got[0], got[1], "text batch issued while others were pending")
    finally:
        faulthandler.cancel_dump_traceback_later()
