#!/usr/bin/env python3
"""bench.py — query patterns/sec of the match() hot path on MI355X (BASELINE.json metric).

One "step" = one call of the C-ABI batch entry point (`fmx_match_batch`, include/fmx.h) over one batch of synthetic
query patterns: host buffers in, host buffers out.  That is the phase the reference CLI brackets with
TICK("Matching") ... TICK("Done") (cli/src/FuzzyMatch-cli.cc:450-467): H2D of the queries, every kernel,
D2H of the records.  The buffers are pinned (fmx_host_alloc) and the library cuts a batch into passes whose copies
overlap the neighbouring passes' kernels; like the reference's `match()`, the entry point is re-entrant and is
driven from two host threads so that one call's last copy overlaps the other call's first kernels.
`value` = queries of all ranks x K / wall time of K steps (barrier + device sync on both sides, max over ranks).
The same batch run from HBM-resident buffers (device entry point, no PCIe) is reported as `device_resident`, with the
per-stage HIP-event times the roofline objects are priced with.

Workload = BASELINE config C3 (5 M-sentence synthetic corpus, mean 25 tokens, 1 M queries per step, fuzzy 0.7,
nmatch 5) unless overridden.  N > 1 (one process per GPU, torch.distributed / RCCL): the index is replicated, the
queries are sharded by rank and the top-N records are gathered to rank 0 over xGMI (the path's only collective,
SURVEY §8e).  `--scaling weak` (default): 1 M queries per rank per step; `--scaling strong`: BASELINE's shape, the
same 1 M queries cut into N contiguous blocks.  Both are measured in one run when N > 1 (`scaling_other`).

Prints ONE JSON line on rank 0.
"""
import argparse
import concurrent.futures as cf
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# hardware queues for the HIP streams of the calls in flight (libfmx.so sets the same default when it is loaded; here before
# torch makes the process's first HIP call — see fmx_capi.cc, fmx_runtime_defaults)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (sentences, mean_len, queries_per_step, params)
    "C1": (10_000, 25, 1_000, dict(fuzzy=0.8, nmatch=5)),
    "C2": (1_000_000, 20, 100_000, dict(fuzzy=0.7, nmatch=5)),
    "C3": (5_000_000, 25, 1_000_000, dict(fuzzy=0.7, nmatch=5)),
    "C4": (5_000_000, 25, 1_000_000, dict(fuzzy=0.5, nmatch=5, min_subseq_ratio=0.5, idf_penalty=1.0)),
    "C5": (5_000_000, 25, 1_000_000, dict(fuzzy=0.7, nmatch=5, contrast=0.3, contrast_buffer=32)),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CALIBRATION = os.path.join(ROOT, "profiles", "r02_fetch_size_calibration.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=list(CONFIGS))
    ap.add_argument("--sentences", type=int, default=0)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--threads", type=int, default=int(os.environ.get("FMX_BENCH_THREADS", "6")), help="synchronous fmx_match_batch calls kept in flight (one host thread each)")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("FMX_BENCH_HOST_THREADS", "4")),
                    help="calls in flight of the host-buffer legs (PCIe-inclusive, from token strings): each adds two copy streams to its two "
                         "compute streams, and beyond ~24 busy hardware queues the GPU's scheduler starts time-slicing them")
    ap.add_argument("--force-gather", action="store_true", help="N = 1 test of the N > 1 step: records also kept in HBM for the (empty) gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-text-leg", action="store_true", help="skip the leg that starts from token strings")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--child", action="store_true", help="profiling child: device-resident steps only, no output line")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--verify", type=int, default=2000, help="queries of rank 0 checked against the oracle")
    ap.add_argument("--model-queries", type=int, default=20000, help="queries the reference byte model is counted on")
    return ap.parse_args()


def pmc_pass(counter, argv_child, log):
    """One `rocprofv3 --pmc <counter>` pass over a child run of this script (device-resident steps only).
    Returns {kernel name: mean counter value per dispatch over the child's timed dispatches} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix=f"fmx_pmc_{counter}_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--child"] + argv_child
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log(f"[bench] pmc pass {counter} failed (rc {r.returncode}): {r.stderr[-1500:]}")
            return None
        by = {}
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != counter:
                continue
            by.setdefault(row["Kernel_Name"].split("(")[0], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        res = {}
        for k, v in by.items():
            v.sort()
            res[k] = v
        return res
    except Exception as e:  # noqa: BLE001
        log(f"[bench] pmc pass {counter} failed: {e}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch  # plumbing only: device tensors for the device entry point, torch.distributed
    import torch.distributed as dist
    import fmx_import
    fmx = fmx_import.load().fmx
    shard = fmx_import.load().shard
    import synth

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # FMX_BENCH_BACKEND=gloo: dry run of the N > 1 control flow on a box with ONE GPU (every rank on device 0, the collectives
    # over gloo through host memory) — for the builder's tests only; the driver's runs use RCCL, one GPU per rank
    backend = os.environ.get("FMX_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if backend == "gloo" else dev   # where the small collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_sent, mean_len, nq_cfg, pkw = CONFIGS[args.config]
    n_sent = args.sentences or n_sent
    nq_cfg = args.queries or nq_cfg
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    # ---- synthetic corpus / queries (SURVEY §8d generator, same bytes on every box) ----
    t0 = time.time()
    L = synth.lib()
    hc = L.fms_corpus_new(n_sent, float(mean_len), synth.VOCAB_TYPES, synth.CORPUS_SEED)
    c_off, c_tok = synth._take(hc)
    vs = L.fms_vocab_size(hc)

    def queries(seed, n):
        hq = L.fms_queries_new(hc, n, float(mean_len), synth.VOCAB_TYPES, seed)
        off, tok = synth._take(hq)
        L.fms_free(hq)
        return off, tok

    # weak: every rank draws its own nq_cfg queries; strong: all ranks draw the SAME nq_cfg queries and take a block
    w_off, w_tok = queries(synth.QUERY_SEED + rank, nq_cfg)
    s_off_all, s_tok_all = queries(synth.QUERY_SEED, nq_cfg) if world > 1 else (w_off, w_tok)
    L.fms_free(hc)
    s_off, s_tok, _, _ = shard.shard_batch(s_off_all, s_tok_all, rank, world)
    log(f"[bench] generated {n_sent} sentences / {nq_cfg} queries in {time.time() - t0:.1f}s")

    # ---- index: sorted on the GPU, derived, uploaded (not in the metric) ----
    t0 = time.time()
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(local_rank)
    ix.finalize()
    ix.to_device(local_rank)
    ix.set_option("contexts", max(2, args.threads))
    for opt in ("FMX_SUB_QUERIES", "FMX_HOST_SUB_QUERIES", "FMX_HEAVY_MIN", "FMX_SERIAL", "FMX_LIGHT_MAX", "FMX_HEAVY_GRID", "FMX_EXPAND_ORDER", "FMX_B_GRID", "FMX_TURNS", "FMX_DEVICE_TURNS", "FMX_PAIR_FINISH"):
        if os.environ.get(opt):
            ix.set_option(opt[4:].lower(), int(os.environ[opt]))
    info = ix.info()
    log(f"[bench] index built+uploaded in {time.time() - t0:.1f}s: {info.n_suffixes} suffixes, "
        f"{info.device_bytes / 1e9:.2f} GB in HBM")

    params = fmx.Params(**pkw)
    cap = params.nmatch or 16

    class Workload:
        """One batch in pinned host memory (+ a device copy for the device-resident measurement)."""

        def __init__(self, q_off, q_tok):
            self.nq = len(q_off) - 1
            self.q_off = fmx.pinned_copy(q_off.astype(np.uint32))
            self.q_tok = fmx.pinned_copy(q_tok.astype(np.uint32))
            self.total_tokens = int(q_off[-1])
            self.max_len = int(np.diff(q_off).max()) if self.nq else 0
            # one result slot per host thread
            self.out = [(fmx.pinned_empty((max(self.nq, 1), cap), fmx.MATCH_DTYPE), fmx.pinned_empty(max(self.nq, 1), np.uint32),
                         fmx.pinned_empty(max(self.nq, 1), np.uint32)) for _ in range(max(args.threads, 1))]
            self.d_off = self.d_tok = None

        def to_device(self):
            if self.d_off is None:
                self.d_off = torch.from_numpy(self.q_off.astype(np.int32)).to(dev)
                self.d_tok = torch.from_numpy(self.q_tok.astype(np.int32)).to(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the host threads that keep calls in flight live for the whole run: a new thread pays for its HIP thread state on its
    # first call (milliseconds), which must not land inside the timed region
    pool = cf.ThreadPoolExecutor(max(args.threads, 1))

    def run_host_steps(wl, k_steps, gather=None):
        """K calls of fmx_match_batch (pinned host buffers in and out), args.threads of them in flight at a time: step
        i + T is submitted when step i has returned, so the copies at both ends of one call overlap the kernels of the
        other (the calls themselves are synchronous: a caller with a queue of batches would do exactly this).  N > 1:
        each step's records also stay in HBM (fmx_match_batch_keep) and are gathered to rank 0 over RCCL; the gathers
        are launched by this thread in step order, so every rank issues its collectives in the same order."""
        T = max(min(args.host_threads, args.threads), 1)
        stats = []

        def one(i, keep):
            return ix.match_batch(params, wl.q_off, wl.q_tok, cap=cap, want_stats=True, out=wl.out[i % T], keep=keep)[3]

        if True:
            futs = {}

            def submit(i):
                keep = None
                if gather is not None:
                    d_out, d_cnt = gather.buffers(i, host_sync=True)  # the slot's previous gather is over
                    keep = (d_out.data_ptr(), d_cnt.data_ptr())
                futs[i] = pool.submit(one, i, keep)

            for i in range(min(T, k_steps)):
                submit(i)
            for i in range(k_steps):
                stats.append(futs.pop(i).result())
                if gather is not None:
                    gather.launch(i)
                if i + T < k_steps:
                    submit(i + T)
        if gather is not None:
            gather.wait_all()
        return stats

    def timed(fn, k_steps):
        sync()
        t1 = time.perf_counter()
        res = fn(k_steps)
        sync()
        el = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    dev_slots = {}

    def run_device_pipeline(wl, k_steps, gather=None):
        """THE METRIC: K calls of the device entry point — query offsets / tokens already in HBM when the timed region starts,
        records written to HBM — with args.threads calls in flight: ONE host thread enqueues them (fmx_match_batch_device_begin
        returns once a call's launches are queued, _end waits for it), every call works in its own context and the library
        chains them on the GPU so that a call's HBM-bound probe runs beside its predecessor's issue-bound second half.
        No per-stage events, no statistics.  Step i + T is enqueued when step i has been waited for."""
        wl.to_device()
        T = max(args.threads, 1)
        if wl not in dev_slots:
            dev_slots[wl] = [(None, torch.zeros((max(wl.nq, 1), cap, 4), dtype=torch.int32, device=dev),
                              torch.zeros((max(wl.nq, 1),), dtype=torch.int32, device=dev)) for _ in range(T)]
            torch.cuda.synchronize()
        slots = dev_slots[wl]

        def begin(i):
            if gather is not None:   # N > 1: the records are written where the RCCL gather of this step picks them up
                d_out, d_cnt = gather.buffers(i, host_sync=True)  # the slot's previous gather is over
            else:
                _, d_out, d_cnt = slots[i % T]
            return ix.device_begin(params, wl.nq, wl.d_off.data_ptr(), wl.d_tok.data_ptr(), wl.total_tokens, d_out.data_ptr(),
                                   d_cnt.data_ptr(), 0, cap, stream=0, max_len=wl.max_len)

        pending = {i: begin(i) for i in range(min(T, k_steps))}
        for i in range(k_steps):
            ix.device_end(pending.pop(i))
            if gather is not None:
                gather.launch(i)     # the path's one collective: top-N records to rank 0 over xGMI, under the steps in flight
            if i + T < k_steps:
                pending[i + T] = begin(i + T)
        if gather is not None:
            gather.wait_all()
        return slots

    def run_device_steps(wl, k_steps):
        """The same batch, one call at a time with per-stage HIP events (stage times for the rooflines)."""
        wl.to_device()
        d_out = torch.zeros((max(wl.nq, 1), cap, 4), dtype=torch.int32, device=dev)
        d_cnt = torch.zeros((max(wl.nq, 1),), dtype=torch.int32, device=dev)
        sts = []
        for _ in range(k_steps):
            sts.append(ix.match_batch_device(params, wl.nq, wl.d_off.data_ptr(), wl.d_tok.data_ptr(), wl.total_tokens,
                                             d_out.data_ptr(), d_cnt.data_ptr(), 0, cap, stream=0, max_len=wl.max_len))
        return sts, d_out, d_cnt

    weak = Workload(w_off, w_tok)
    strong = Workload(s_off, s_tok) if world > 1 else weak
    main_wl, other_wl = (weak, strong) if args.scaling == "weak" else (strong, weak)

    if args.child and os.environ.get("FMX_CHILD_MODE") == "pipeline":  # design-time: trace of the timed region itself
        run_device_pipeline(main_wl, 2 * max(args.threads, 1) + args.warmup)
        torch.cuda.synchronize()
        run_device_pipeline(main_wl, args.steps)
        torch.cuda.synchronize()
        return
    if args.child:  # profiling child of a --pmc pass: device-resident steps, nothing else
        run_device_steps(main_wl, args.warmup)
        torch.cuda.synchronize()
        run_device_steps(main_wl, args.steps)
        torch.cuda.synchronize()
        return

    # ---- PCIe-inclusive steps through fmx_match_batch (reported beside the metric, never as `value`) ----
    def measure(wl):
        gather = shard.RecordGather(wl.nq, cap, dev, dst=0, depth=max(args.threads, 1) + 1, coll_device=cdev) if (world > 1 or args.force_gather) else None
        # set-up, not warm-up: every call in flight works in its own context, and a context sizes its workspace on its
        # first batches (allocation + re-runs); two rounds of T concurrent calls leave all T contexts grown
        run_host_steps(wl, 2 * max(args.threads, 1), gather)
        run_host_steps(wl, args.warmup, gather)
        el, sts = timed(lambda k: run_host_steps(wl, k, gather), args.steps)
        return el, sts

    def measure_device(wl):
        gather = shard.RecordGather(wl.nq, cap, dev, dst=0, depth=max(args.threads, 1) + 1, coll_device=cdev) if (world > 1 or args.force_gather) else None
        run_device_pipeline(wl, 2 * max(args.threads, 1), gather)   # set-up: every context in flight sizes its workspace once
        run_device_pipeline(wl, args.warmup, gather)
        el, _ = timed(lambda k: run_device_pipeline(wl, k, gather), args.steps)
        if gather is not None:  # the parity check below reads slot 0 of the plain buffers
            run_device_pipeline(wl, 1)
        return el

    def all_ranks(n):
        if world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        return int(t.item())

    def measure_from_strings(wl):
        """SURVEY §8(d)'s bracket includes the per-query host prep — VocabIndexer::getIndex per pattern token (src/fuzzy_match.cc:
        468-470).  This leg starts from token STRINGS: every step looks its tokens up (fmx_vocab_lookup_mt, host threads) into the
        pinned id buffer and then runs the host-buffer call.  The spellings are the generator's (tools/fm_synth.c: fms_word)."""
        import ctypes
        buf = ctypes.create_string_buffer(32)
        words = [b"", b""]
        for k in range(vs - 2):
            n = L.fms_word(k, buf, 32)
            words.append(buf.raw[:n])
        offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
        pos = 0
        for k in range(2, len(words)):
            offs[k] = pos
            pos += len(words[k]) + 1
        arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
        word_ptrs = ix.set_forms_packed(arena, offs)
        tok_ptrs = np.ascontiguousarray(word_ptrs[wl.q_tok])          # one char pointer per pattern token of the batch
        T = max(min(args.host_threads, args.threads), 1)
        cores = max(1, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
        try:
            import oracle_lib as _ol
            cores = _ol.effective_cpus()
        except Exception:  # noqa: BLE001
            pass
        per_call = max(1, cores // T)
        ids = [fmx.pinned_empty(len(wl.q_tok), np.uint32) for _ in range(T)]
        t_lookup = [0.0]

        def one(i):
            a = time.perf_counter()
            ix.vocab_lookup_ptrs(tok_ptrs, ids[i % T], per_call)
            t_lookup[0] += time.perf_counter() - a
            return ix.match_batch(params, wl.q_off, ids[i % T], cap=cap, want_stats=True, out=wl.out[i % T])[3]

        def run(k_steps):
            futs = {i: pool.submit(one, i) for i in range(min(T, k_steps))}
            for i in range(k_steps):
                futs.pop(i).result()
                if i + T < k_steps:
                    futs[i + T] = pool.submit(one, i + T)

        run(T)
        assert np.array_equal(ids[0], wl.q_tok), "vocabulary lookup does not reproduce the ids"
        t_lookup[0] = 0.0
        k = max(6, args.steps // 2)
        el, _ = timed(run, k)
        return {"value": round(wl.nq * k / el, 1), "ms_per_step": round(el / k * 1e3, 3), "steps": k,
                "lookup_ms_per_step": round(t_lookup[0] / k * 1e3, 3), "lookup_threads_per_call": per_call, "calls_in_flight": T,
                "tokens_per_step": int(len(wl.q_tok)),
                "note": "fmx_vocab_lookup_mt (token strings -> ids, in the timed region) + fmx_match_batch (pinned host buffers)"}

    # The metric first.  (Round 3, first half: the host-buffer steps ran first because at C2 they lost a third when measured after
    # the device-resident ones; with eight hardware queues the order no longer matters to them — 1.41-1.53 ms either way — while
    # a host-buffer leg with too many busy queues can leave the process's queues time-sliced and cost the metric 5 %.)
    host_first = os.environ.get("FMX_BENCH_ORDER", "dev_first") == "host_first"
    if host_first:
        elapsed_host, host_stats = measure(main_wl)
    # ---- THE METRIC: inputs resident in HBM when the timed region starts ----
    elapsed = measure_device(main_wl)
    nq_step_all = all_ranks(main_wl.nq)
    value = nq_step_all * args.steps / elapsed
    # ---- the same steps from pinned host buffers (H2D + kernels + D2H; + the RCCL gather of the records when N > 1) ----
    if not host_first:
        elapsed_host, host_stats = measure(main_wl)
    value_host = nq_step_all * args.steps / elapsed_host
    scaling_other = None
    if world > 1:
        el2 = measure_device(other_wl)
        el2h, _ = measure(other_wl)
        n2 = all_ranks(other_wl.nq)
        scaling_other = {"scaling": "strong" if args.scaling == "weak" else "weak", "value": round(n2 * args.steps / el2, 1),
                         "ms_per_step": round(el2 / args.steps * 1e3, 3), "queries_per_step_all_ranks": n2,
                         "pcie_inclusive_value": round(n2 * args.steps / el2h, 1)}

    from_strings = measure_from_strings(main_wl) if (world == 1 and not args.no_text_leg) else None

    def measure_from_packed_text(wl):
        """The same bracket with the id lookup ON THE DEVICE (fmx_match_batch_text): a step hands over its patterns as packed token
        spellings — NUL-terminated, back to back, what a tokenizer writes — in pinned host memory; the spellings cross PCIe instead
        of the ids, k_vocab_lookup turns them into ids in HBM, the pipeline follows.  Records come back to pinned host buffers."""
        import ctypes
        buf = ctypes.create_string_buffer(32)
        wlen = np.zeros(vs, np.int64)
        parts = []
        woff = np.zeros(vs, np.int64)
        pos = 0
        for k in range(vs - 2):
            n = L.fms_word(k, buf, 32)
            parts.append(buf.raw[:n] + b"\0")
            wlen[k + 2] = n + 1
            woff[k + 2] = pos
            pos += n + 1
        arena = np.frombuffer(b"".join(parts), dtype=np.uint8)
        tl = wlen[wl.q_tok]
        total = int(tl.sum())
        text = fmx.pinned_empty(total, np.uint8)
        at = 0
        for lo in range(0, len(wl.q_tok), 1 << 21):  # in slices: the gather index of the whole batch would be gigabytes
            t = wl.q_tok[lo:lo + (1 << 21)]
            ln = wlen[t]
            n = int(ln.sum())
            start = np.cumsum(ln) - ln
            idx = np.repeat(woff[t] - start, ln) + np.arange(n)
            text[at:at + n] = arena[idx]
            at += n
        T = max(min(args.host_threads, args.threads), 1)

        def one(i):
            return ix.match_batch_text(params, wl.q_off, text, cap=cap, want_stats=True, out=wl.out[i % T])[3]

        def run(k_steps):
            futs = {i: pool.submit(one, i) for i in range(min(T, k_steps))}
            for i in range(k_steps):
                futs.pop(i).result()
                if i + T < k_steps:
                    futs[i + T] = pool.submit(one, i + T)

        run(2 * T)
        ref = ix.match_batch(params, wl.q_off, wl.q_tok, cap=cap)
        got = ix.match_batch_text(params, wl.q_off, text, cap=cap)
        assert np.array_equal(ref[1], got[1]) and ref[0][:2000].tobytes() == got[0][:2000].tobytes(), "text batch differs from the id batch"
        k = max(6, args.steps // 2)
        el, _ = timed(run, k)
        return {"value": round(wl.nq * k / el, 1), "ms_per_step": round(el / k * 1e3, 3), "steps": k, "calls_in_flight": T,
                "text_bytes_per_step": total, "tokens_per_step": int(len(wl.q_tok)),
                "note": "fmx_match_batch_text: packed token spellings in pinned host memory -> H2D -> k_vocab_lookup (ids in HBM) -> pipeline -> "
                        "records to pinned host buffers; checked against the id batch"}

    from_text = None
    if world == 1 and not args.no_text_leg:
        try:
            from_text = measure_from_packed_text(main_wl)
        except Exception as e:  # noqa: BLE001  (a side leg: the line is still worth printing)
            log(f"[bench] from_packed_text leg failed: {e!r}")

    # ---- the same batch from HBM (device entry point): stage times for the rooflines ----
    run_device_steps(main_wl, 1)
    el_dev, (dev_stats, d_out, d_cnt) = timed(lambda k: run_device_steps(main_wl, k), args.steps)
    names = fmx.stage_names()
    stage_ms = np.zeros(16)
    for st in dev_stats:
        stage_ms += np.array(list(st.ms_kernel))
    per_stage = {n: float(stage_ms[i] / max(args.steps, 1)) for i, n in enumerate(names)}
    last = dev_stats[-1]
    # per-rank probe GB/s of its own byte model needs the counting variant of the kernel: one untimed step
    ix.set_option("probe_count", 1)
    (cst,), _, _ = run_device_steps(main_wl, 1)
    ix.set_option("probe_count", 0)
    torch.cuda.synchronize()

    # stage "expand_a" = k_expand_a and k_expand_h side by side (main stream: until both are done); "expand_l" = the second
    # scan + k_expand_b<true> with k_expand_b<false> beside it, "expand_wait" = what is left of the latter afterwards
    probe_ms, exa_ms = per_stage.get("probe", 0.0), per_stage.get("expand_a", 0.0)
    # byte models of THIS algorithm, per launch, from the kernels' own counters:
    #   k_probe     bytes its loads request (n-gram, qva pair, 16 B table entries, 32 B keys per bisection level, ...)
    #   expand_a    k_expand_a: 8 B per hit it walks (short ranges of the queries that are not light); k_expand_h: 8 B per
    #               plane word + 8 B per hit re-tested exactly; both: 8 B per hit queued for k_expand_b<false>; 8 B per
    #               (query, start) range descriptor of the queries they visit (upper bound: all)
    b_probe = float(cst.probe_bytes)
    walked_a = last.n_hits - last.n_heavy_hits - last.n_light_hits
    queued = last.n_phase_b - last.n_light_hits
    b_exa = 8.0 * walked_a + 8.0 * last.plane_words + 8.0 * last.n_heavy_pass + 8.0 * queued + 8.0 * main_wl.total_tokens
    #   expand_b    k_expand_b<true> (light queries): 8 B range descriptor per start position + 16 B record per hit;
    #               k_expand_b<false>: 8 B queue entry + 16 B record per queued hit; both: the pattern once per query, and for
    #               every hit kept the sentence's tokens (mean length of the corpus) and a 16 B segment entry; the tokens of
    #               hits that fail the dominance / coverage tests are not counted (a lower bound of what the kernels read)
    exb_ms = per_stage.get("expand_l", 0.0) + per_stage.get("expand_wait", 0.0)
    b_exb = (16.0 * last.n_light_hits + 24.0 * queued + 12.0 * main_wl.total_tokens
             + last.n_kept * (4.0 * (mean_len + 1) + 16.0))
    probe_gbs = b_probe / (probe_ms * 1e-3) / 1e9 if probe_ms > 0 else 0.0
    exa_gbs = b_exa / (exa_ms * 1e-3) / 1e9 if exa_ms > 0 else 0.0
    rank_rates = None
    if world > 1:
        t = torch.tensor([probe_gbs, exa_gbs, per_stage.get("probe", 0.0), el_dev / args.steps * 1e3], dtype=torch.float64, device=cdev)
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)
        rank_rates = [{"rank": r, "probe_model_gbps": round(float(x[0]), 1), "expand_a_model_gbps": round(float(x[1]), 1),
                       "probe_ms": round(float(x[2]), 4), "device_ms_per_step": round(float(x[3]), 3)} for r, x in enumerate(g)]

    line = None
    if rank == 0:
        log(f"[bench] value (inputs in HBM, {args.threads} calls in flight): {value / 1e6:.2f} M patterns/s, {elapsed / args.steps * 1e3:.3f} ms/step; "
            f"PCIe-inclusive: {value_host / 1e6:.2f} M patterns/s, {elapsed_host / args.steps * 1e3:.3f} ms/step "
            f"(pinned={host_stats[-1].pinned}); one call at a time with stage events: {main_wl.nq * args.steps / el_dev / 1e6:.2f} M patterns/s")
        log(f"[bench] host-path calls: retries {sum(int(s.n_retries) for s in host_stats)}, ms per call min/median/max "
            f"{min(s.ms_total for s in host_stats):.2f}/{float(np.median([s.ms_total for s in host_stats])):.2f}/{max(s.ms_total for s in host_stats):.2f}")
        log(f"[bench] kernel ms/step by stage: { {k: round(v, 3) for k, v in per_stage.items()} }")
        log(f"[bench] last step: hits {last.n_hits} (bit planes {last.n_heavy_hits}, passed {last.n_heavy_pass}) phaseB {last.n_phase_b} "
            f"(light queries {last.n_light_hits}) kept {last.n_kept} candidates {last.n_candidates} retries {last.n_retries} big {last.n_big} max_kept {last.max_kept}")

        # ---- HBM traffic of the two kernels, measured now: rocprofv3 --pmc child passes (separate passes) ----
        traffic = {}
        if world == 1 and not args.no_pmc:
            child = ["--config", args.config, "--steps", "2", "--warmup", "1", "--scaling", args.scaling]
            if args.sentences:
                child += ["--sentences", str(args.sentences)]
            if args.queries:
                child += ["--queries", str(args.queries)]
            cal = json.load(open(CALIBRATION)) if os.path.exists(CALIBRATION) else {}
            t1 = time.time()
            fetch = pmc_pass("FETCH_SIZE", child, log)
            write = pmc_pass("WRITE_SIZE", child, log) if fetch else None
            log(f"[bench] pmc child passes: {time.time() - t1:.0f}s")
            if fetch and write:
                rounds = int(pkw.get("nmatch", 1)) if pkw.get("contrast", 0.0) > 0.0 else 1
                for kern, pattern, per_step in (("fmx::k_probe", "random", 1), ("fmx::k_expand_a", "stream16", 1), ("fmx::k_expand_h", "stream8", 1),
                                                ("fmx::k_expand_b", "random", 1), ("fmx::k_dp_reg", "random", 1), ("fmx::k_finish", "random", 1),
                                                ("fmx::k_camap", "random", 1), ("fmx::k_cround", "random", rounds)):
                    keys = [k for k in fetch if k.replace("void ", "").startswith(kern)]  # template instantiations add up
                    if not keys:
                        continue
                    # the child's two timed steps (per_step dispatches each): bytes per step
                    f_kb = sum(np.sum([v for _, v in fetch[k][-2 * per_step:]]) / 2.0 for k in keys)
                    w_kb = sum(np.sum([v for _, v in write.get(k, [(0, 0.0)])[-2 * per_step:]]) / 2.0 for k in keys)
                    f_fac = float(cal.get("fetch_factor", {}).get(pattern, 2.0))
                    w_fac = float(cal.get("write_factor", 1.0))
                    traffic[kern.split("::")[1]] = {"bytes": int(f_kb * 1024 * f_fac + w_kb * 1024 * w_fac), "fetch_kb_raw": float(f_kb),
                                                    "write_kb_raw": float(w_kb), "fetch_factor": f_fac, "write_factor": w_fac,
                                                    "calibrated": bool(cal)}

        def roof(kernel, model_bytes, ms, tr_keys, what):
            ach = model_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr = sum(traffic[k]["bytes"] for k in tr_keys if k in traffic) if all(k in traffic for k in tr_keys) and tr_keys else None
            return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr,
                    "traffic_gbps": round(tr / (ms * 1e-3) / 1e9, 1) if tr and ms > 0 else None,
                    "traffic_frac": round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr and ms > 0 else None,
                    "algorithmic_bytes_per_launch": int(model_bytes), "kernel_ms": round(ms, 4), "launches_per_step": 1,
                    "model": what}

        r_probe = roof("k_probe", b_probe, probe_ms, ["k_probe"],
                       "bytes requested by the kernel's own loads, counted by k_probe<COUNT> on the same batch: seed n-gram, "
                       "_quickVocabAccess pair, 16 B trigram / 5-gram entries, two 32 B keys per bisection level")
        r_exa = roof("k_expand_h+k_expand_a", b_exa, exa_ms, ["k_expand_a", "k_expand_h"],
                     "phase A of the queries with long ranges or many hits (two kernels side by side): 8 B per bit-plane word, "
                     "8 B per plane survivor re-tested, 8 B per hit walked one by one, 8 B per hit queued, 8 B per range "
                     "descriptor (kernel counters)")
        r_probe["lines_bytes_per_launch"] = int(cst.probe_lines) * 128  # distinct 128 B lines x 128: what random reads cost the memory system
        r_exb = roof("k_expand_b<true>+k_expand_b<false>", b_exb, exb_ms, ["k_expand_b"],
                     "hit walk of the light queries + phase B of the others (side by side): 8 B range descriptor per start, 16 B "
                     "record per light hit, 8 B queue entry + 16 B record per queued hit, 4 B per pattern token, and per kept "
                     "hit its sentence (4 B x (mean length + 1)) and a 16 B segment entry; reads for hits that fail the "
                     "dominance / exact coverage tests are not counted (lower bound)")
        # edit distances: per candidate its sentence (length word + tokens), the 16 B candidate record read and written back, the
        # 8 B list entry; the pattern is shared by the candidates of a query (4 B per pattern token per query)
        dp_ms = per_stage.get("dp", 0.0)
        b_dp = last.n_candidates * (4.0 * (mean_len + 1) + 16.0 + 16.0 + 8.0) + 4.0 * main_wl.total_tokens
        r_dp = roof("k_dp_reg<32|64>", b_dp, dp_ms, ["k_dp_reg"],
                    "one lane per (query, candidate): 4 B x (sentence length + 1) of the sentence, 16 B candidate record in and out, 8 B "
                    "list entry, 4 B per pattern token per query (kernel counters; mean sentence length of the corpus)")
        roofs = sorted([r_probe, r_exa, r_exb], key=lambda r: -r["kernel_ms"]) + [r_dp]
        if pkw.get("contrast", 0.0) > 0.0:
            # the contrastive rerank: per sentence-pair distance both sentences and the per-match state (list entry 8, record 16,
            # keys 8 + 8, penalty sum 4 + 4, penalty 4, two sentence positions 8); per candidate the 16 B record the replay reads
            fin_ms = per_stage.get("finish", 0.0)
            b_fin = last.n_rerank * (8.0 * (mean_len + 1) + 64.0) + 16.0 * last.n_candidates
            roofs.append(roof("finish stage: k_finish<64> + k_camap + nmatch x k_cround<32|64>", b_fin, fin_ms, ["k_finish", "k_camap", "k_cround"],
                              f"replay + contrastive rerank; {int(last.n_rerank)} sentence-pair edit distances per step (kernel counter) x (two sentences "
                              "of 4 B x (mean length + 1) + 64 B of per-match state) + 16 B per candidate; the rounds are bound by VALU issue "
                              "(~1800 instructions per distance, VALU busy 0.75: profiles/r03_c5_pmc_sq.csv), not by HBM — priced here because the "
                              "line's schema has no issue roofline"))
        dominant, others = roofs[0], roofs[1:]

        # ---- parity spot check, reference byte model, CPU baseline (the oracle: test infrastructure, outside the timed region) ----
        import oracle_lib
        t1 = time.time()
        o = oracle_lib.OracleIndex()
        o.set_arrays(ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos"),
                     ix.array("qva"), ix.array("sfreq"))
        log(f"[bench] oracle index ready in {time.time() - t1:.1f}s")
        oprm = oracle_lib.Params(**pkw)
        q_off, q_tok = main_wl.q_off, main_wl.q_tok
        nq = main_wl.nq
        parity = ref_model = cpu_baseline = None
        nv = min(args.verify, nq)
        if nv:
            oo, oc = o.match_batch(oprm, q_off[:nv + 1], q_tok[:int(q_off[nv])], cap, nthreads=oracle_lib.effective_cpus())
            bad_h = bad_d = bad_p = 0
            ho, hc_, _ = main_wl.out[0]
            go = d_out[:nv].cpu().numpy().view(oracle_lib.MATCH_DTYPE).reshape(nv, cap)
            gc = d_cnt[:nv].cpu().numpy().astype(np.uint32)
            _, p_out, p_cnt = dev_slots[main_wl][0]          # what the timed (pipelined, stats-less) device calls left in HBM
            po = p_out[:nv].cpu().numpy().view(oracle_lib.MATCH_DTYPE).reshape(nv, cap)
            pc = p_cnt[:nv].cpu().numpy().astype(np.uint32)
            for q in range(nv):
                ref = oo[q, :oc[q]].tobytes()
                bad_h += int(oc[q] != hc_[q] or ref != ho[q, :hc_[q]].tobytes())
                bad_d += int(oc[q] != gc[q] or ref != go[q, :gc[q]].tobytes())
                bad_p += int(oc[q] != pc[q] or ref != po[q, :pc[q]].tobytes())
            parity = {"checked": int(nv), "mismatches": int(bad_h + bad_d + bad_p),
                      "paths": "the timed device calls, the host-buffer entry point and the staged device calls vs oracle"}
        nm = min(args.model_queries, nq)
        if nm:
            _, _, ctr = o.match_batch(oprm, q_off[:nm + 1], q_tok[:int(q_off[nm])], cap, nthreads=oracle_lib.effective_cpus(),
                                      want_counters=True)
            d = ctr.as_dict()
            nqc = max(d["queries"], 1)
            ref_model = {"queries_counted": int(nqc), "B_probe_per_query": round(d["probe_bytes"] / nqc, 1),
                         "B_reg_per_query": round(d.get("reg_bytes", 0) / nqc, 1), "B_cand_per_query": round(d.get("cand_bytes", 0) / nqc, 1),
                         "note": "SURVEY §8d's model of the REFERENCE algorithm's logical reads, counted by the oracle; the hash tables and "
                                 "bit planes skip most of that work, so pricing this build on it gives fractions above 1 "
                                 f"(probe: {d['probe_bytes'] / nqc * nq / max(probe_ms, 1e-9) / 1e6 / HBM_PEAK_GBS:.2f})"}
        if not args.no_cpu_baseline and world == 1:
            cores = oracle_lib.effective_cpus()  # cgroup CPU quota, not the hardware threads of the host
            ns = min(nq, max(2000, 64 * cores))
            t1 = time.perf_counter()
            o.match_batch(oprm, q_off[:ns + 1], q_tok[:int(q_off[ns])], cap, nthreads=cores)
            rate = ns / (time.perf_counter() - t1)
            ns2 = int(min(nq, max(ns, rate * args.cpu_seconds)))
            t1 = time.perf_counter()
            o.match_batch(oprm, q_off[:ns2 + 1], q_tok[:int(q_off[ns2])], cap, nthreads=cores)
            dt = time.perf_counter() - t1
            cpu_baseline = {"value": round(ns2 / dt, 1), "unit": "patterns/s", "cores": cores, "kind": "port",
                            "sample": f"first {ns2} queries of the same {args.config} workload, oracle/fm_oracle.c "
                                      f"fmo_match_batch with {cores} threads (= usable CPUs: min(affinity, cgroup cpu.max) of "
                                      f"{os.cpu_count()} hardware threads), {dt:.1f}s"}
        o.close()

        line = {
            "metric": "query patterns/sec @ fuzzy=0.7, 5M-sentence .fmi; 1/2/4/8 GPU + CPU ref",
            "value": round(value, 1), "unit": "patterns/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_sent}-sentence synthetic corpus (mean {mean_len} tok, near-duplicate families), "
                                   f"{nq_cfg} queries per {'GPU' if args.scaling == 'weak' else 'job'} per step, "
                                   + " ".join(f"{k}={v}" for k, v in pkw.items()),
                       "queries_per_step_all_ranks": nq_step_all, "sentences": n_sent, "index_bytes_hbm": int(info.device_bytes),
                       "timed_path": "fmx_match_batch_device_begin / _end: query offsets / tokens resident in HBM when the timed region starts, "
                                     f"records written to HBM, {args.threads} call(s) in flight enqueued by one host thread (each in its own context)"
                                     + (", + RCCL gather of every step's records to rank 0 (asynchronous, under the steps in flight)" if world > 1 else ""),
                       "parallelism": f"queries sharded over {world} GPU(s), index replicated"},
            "roofline": dominant, "roofline_other": others, "reference_byte_model": ref_model,
            "pcie_inclusive": {"value": round(value_host, 1), "ms_per_step": round(elapsed_host / args.steps * 1e3, 3),
                               "note": "the same steps through fmx_match_batch: pinned host buffers in and out (H2D + kernels + D2H"
                                       + (" + the RCCL gather of the records to rank 0" if world > 1 else "") + f"), {max(min(args.host_threads, args.threads), 1)} call(s) in flight"},
            "from_token_strings": from_strings, "from_packed_text": from_text,
            "staged": {"value": round(nq_step_all * args.steps / el_dev, 1),
                       "ms_per_step": round(el_dev / args.steps * 1e3, 3),
                       "stage_ms_per_step": {k: round(v, 4) for k, v in per_stage.items()},
                       "note": "same batch from HBM, ONE call at a time, every pass synchronised and timed per stage with HIP events on "
                               "the launch stream (the kernel times the rooflines are priced with)"},
            "scaling_other": scaling_other, "per_rank": rank_rates,
            "cpu_baseline": cpu_baseline, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
