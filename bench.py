#!/usr/bin/env python3
"""bench.py — query patterns/sec of the match() hot path on MI355X (BASELINE.json metric).

One "step" = one batch of synthetic query patterns through the C-ABI (include/fmx.h).  `value` is measured on the bracket
SURVEY §8(d) gives the metric — the phase the reference CLI times between TICK("Matching") and TICK("Done")
(cli/src/FuzzyMatch-cli.cc:450-467): per-query host prep (vocabulary id lookup), H2D of the queries, every kernel, D2H of
the records.  A step of that leg is one text batch: the patterns are handed over as what a tokenizer holds — packed token
SPELLINGS in pinned host memory —, the ids are looked up on the device (k_vocab_lookup), the pipeline runs, the top-N records come
back to pinned host buffers.  One host thread keeps four steps in flight (fmx_match_batch_text_begin / _end, each step in its own
context), like the reference's N worker threads keep N patterns in flight.

Every leg rotates DISTINCT query batches (seeds QUERY_SEED + k, one more batch than calls in flight) through its calls, so that
calls overlapping on the GPU never walk the same index lines a fraction of a millisecond apart.

Side legs in the same JSON line (never `value`):
  text_sync_callers   the value leg's bracket from four synchronous callers (fmx_match_batch_text on as many host threads)
  device_resident     the same steps through fmx_match_batch_device_begin/_end: offsets / ids already in HBM, records left in HBM
  pcie_inclusive_ids  fmx_match_batch: vocabulary ids in pinned host memory in, records out (no lookup: the caller already holds ids)
  from_token_strings  fmx_vocab_lookup_mt on host threads inside the timed region, then fmx_match_batch
  cli_end_to_end      tools/fmx-cli --tokenised on the run's own index (through a .fmi file) and first batch: stdin -> stdout
  cli_raw_text        the same through the CLI's raw-text mode (the reference CLI's only one): lines tokenised on the device
  staged              one device-resident call at a time with per-stage HIP events: the kernel times the rooflines are priced with

Workload = BASELINE config C3 (5 M-sentence synthetic corpus, mean 25 tokens, 1 M queries per step, fuzzy 0.7, nmatch 5) unless
overridden.  N > 1 (one process per GPU, torch.distributed / RCCL): the index is replicated, the queries are sharded by rank and
the top-N records are gathered to rank 0 over xGMI (the path's only collective).  `--scaling strong` (the default when
WORLD_SIZE > 1: BASELINE's shape, one 1 M-query batch cut into N contiguous blocks) or `weak` (1 M queries per rank per step); the
other one is measured too and reported as `scaling_other`.

Prints ONE JSON line on rank 0.
"""
import argparse
import concurrent.futures as cf
import csv
import datetime
import glob
import gc
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
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto = strong when WORLD_SIZE > 1 (BASELINE's C3 is ONE 1 M-query batch cut over 1/2/4/8 GPUs), weak at N = 1 (the same thing there)")
    ap.add_argument("--threads", type=int, default=int(os.environ.get("FMX_BENCH_THREADS", "6")), help="device-resident calls kept in flight (one host thread enqueues them)")
    ap.add_argument("--host-threads", type=int, default=int(os.environ.get("FMX_BENCH_HOST_THREADS", "4")),
                    help="synchronous host-buffer calls in flight (one host thread each) — the side legs `text_sync_callers`, `pcie_inclusive_ids`, "
                         "`from_token_strings` (the library admits at most max_host_active = 4 of them at once whatever this says)")
    ap.add_argument("--text-calls", type=int, default=int(os.environ.get("FMX_BENCH_TEXT_CALLS", "0")),
                    help="text batches the value leg keeps in flight (fmx_match_batch_text_begin / _end, one host thread; the leg raises the "
                         "library's max_host_active to this).  Five since round 5, which traced the 'cliff' of rounds 2-4: five or six calls begun "
                         "back to back on an idle GPU make the FIRST take 13-340 ms (the library now starts host-buffer calls 0.5 ms apart, which "
                         "avoids it: tools/dbg/text_host_time.py); seven and more crawl whatever the stream layout.  Over 20 steps from an idle "
                         "pipeline: four 5.36-5.42 ms per step, five 5.19-5.28, six 5.76 (a longer fill and drain).  0 = five for steps of half a million "
                         "queries and more, four for smaller ones (C2 and the 125 k-query shard: 0.88-0.93 / 1.04 ms with four, 0.97 / 1.08 with five)")
    ap.add_argument("--batches", type=int, default=0, help="distinct query batches rotated through the steps (0 = calls in flight + 1)")
    ap.add_argument("--force-gather", action="store_true", help="N = 1 test of the N > 1 step: records also kept in HBM for the (empty) gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-text-leg", action="store_true", help="skip the side leg that looks the ids up on host threads")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--child", action="store_true", help="profiling child: device-resident steps only, no output line")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--verify", type=int, default=20000, help="queries of rank 0 checked against the oracle in every leg, spread evenly over a batch")
    ap.add_argument("--model-queries", type=int, default=20000, help="queries the reference byte model is counted on")
    ap.add_argument("--no-via-fmi", action="store_true", help="skip the .fmi round trip of the index (the default at N > 1: every rank would write its own 2 GB file)")
    ap.add_argument("--no-default-index", action="store_true", help="skip the cli_default_index leg (N = 1 only): the workload on an index built from cased TEXT "
                    "by fmx-cli -c under the reference's default -p nbr,tag,cas, whose real tokens differ from its normalised tokens")
    ap.add_argument("--via-fmi", action="store_true", help="(the default at N = 1 since round 6) the index goes through the reference's on-disk format first: fmx_index_save -> fmx_index_load "
                                                           "(arrays compared, times and file size reported); the steps then run on the LOADED index")
    return ap.parse_args()


def pmc_pass(counter, argv_child, log):
    """One `rocprofv3 --pmc <counter>` pass over a child run of this script (device-resident steps only).
    Returns {kernel name: [(dispatch id, counter value)] over the child's dispatches} or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix=f"fmx_pmc_{counter}_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if "LOCAL_RANK" in env and "HIP_VISIBLE_DEVICES" not in env and env.get("FMX_BENCH_BACKEND") != "gloo":
        env["HIP_VISIBLE_DEVICES"] = env["LOCAL_RANK"]   # the child is one process on this rank's GPU
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
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


def take_queries(q_off, q_tok, idx):
    """CSR of the queries idx (ascending) of a batch."""
    lens = (q_off[idx + 1] - q_off[idx]).astype(np.int64)
    off = np.zeros(len(idx) + 1, np.uint32)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    start = q_off[idx].astype(np.int64)
    pos = np.repeat(start - off[:-1].astype(np.int64), lens) + np.arange(total, dtype=np.int64)
    return off, q_tok[pos]


def main():
    args = parse_args()
    # a run that is still here after BENCH_WATCHDOG_S seconds (default 25 minutes; a default run takes about four) says where every
    # thread stands and ends with a non-zero exit code instead of waiting for the caller's limit (round 6: one C5 run stopped
    # answering in its host-buffer legs and left no trace; tools/dbg/stress_legs.py is the loop that looked for it, HISTORY.md)
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("BENCH_WATCHDOG_S", "1500")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if args.scaling == "auto":
        args.scaling = "strong" if world > 1 else "weak"

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
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        else:
            # (rank 0 checks parity against the oracle and runs the rocprofv3 child passes while the others wait at the last barrier)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=30))

    n_sent, mean_len, nq_cfg, pkw = CONFIGS[args.config]
    n_sent = args.sentences or n_sent
    nq_cfg = args.queries or nq_cfg
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    T_dev = max(args.threads, 1)
    T_host = max(min(args.host_threads, args.threads), 1)
    nq_rank = nq_cfg // world if (args.scaling == "strong" and world > 1) else nq_cfg   # queries of one step on this rank
    T_text = max(min(args.text_calls or (5 if nq_rank >= (1 << 19) else 4), args.threads), 1)
    NB = args.batches or (T_dev + 1)          # distinct batches of the device-resident legs
    NB_host = min(NB, T_host + 1) if not args.batches else NB   # ... of the host-buffer legs (fewer calls in flight)

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

    # batch k: weak = every rank draws its own nq_cfg queries (seed QUERY_SEED + k + 101 x rank); strong = all ranks draw the SAME
    # nq_cfg queries (seed QUERY_SEED + k) and take their contiguous block
    def draw(k, scaling):
        if scaling == "weak" or world == 1:
            return queries(synth.QUERY_SEED + k + 101 * rank, nq_cfg)
        off_all, tok_all = queries(synth.QUERY_SEED + k, nq_cfg)
        off, tok, _, _ = shard.shard_batch(off_all, tok_all, rank, world)
        return off, tok

    raw_main = [draw(k, args.scaling) for k in range(NB)]
    # N > 1: rank 0 also holds EVERY rank's block of the batch the value leg's last timed step runs, to check what the RCCL gather
    # assembled from the other ranks against the oracle (cli/src/FuzzyMatch-cli.cc:151-162: results in input order)
    n_text_batches = min(NB, T_text + 1)
    b_last_text = (args.steps - 1) % n_text_batches
    all_blocks = None
    if world > 1 and rank == 0 and args.verify:
        if args.scaling == "strong":
            off_all, tok_all = queries(synth.QUERY_SEED + b_last_text, nq_cfg)
            all_blocks = [shard.shard_batch(off_all, tok_all, r, world)[:2] for r in range(world)]
            del off_all, tok_all
        else:
            all_blocks = [queries(synth.QUERY_SEED + b_last_text + 101 * r, nq_cfg) for r in range(world)]
    other_scaling = "strong" if args.scaling == "weak" else "weak"
    raw_other = [draw(k, other_scaling) for k in range(NB_host)] if world > 1 else []
    L.fms_free(hc)
    log(f"[bench] generated {n_sent} sentences, {NB} batches of {nq_cfg} queries in {time.time() - t0:.1f}s")

    # ---- index: sorted on the GPU, derived, uploaded (not in the metric) ----
    t0 = time.time()
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(local_rank)
    ix.finalize()
    # the spellings of the generator's vocabulary (tools/fm_synth.c: fms_word): text batches are looked up against them, and a saved
    # .fmi carries real words
    import ctypes
    wbuf = ctypes.create_string_buffer(32)
    words = [b"", b""]
    for k in range(vs - 2):
        n = L.fms_word(k, wbuf, 32)
        words.append(wbuf.raw[:n])
    w_offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
    pos = 0
    for k in range(2, len(words)):
        w_offs[k] = pos
        pos += len(words[k]) + 1
    w_arena = np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy()
    word_ptrs = ix.set_forms_packed(w_arena, w_offs)
    via_fmi = None
    if (args.via_fmi or world == 1) and not args.no_via_fmi:
        via_fmi = through_fmi(fmx, ix, local_rank, rank, log)
        ix = via_fmi.pop("index")
        # (the loaded index holds the same spellings — they came through the file —: word_ptrs keeps pointing into w_arena)
    else:
        ix.to_device(local_rank)
    ix.set_option("contexts", max(2, T_dev))
    for opt in ("FMX_SUB_QUERIES", "FMX_HOST_SUB_QUERIES", "FMX_HEAVY_MIN", "FMX_SERIAL", "FMX_LIGHT_MAX", "FMX_HEAVY_GRID", "FMX_EXPAND_ORDER", "FMX_B_GRID", "FMX_TURNS", "FMX_DEVICE_TURNS", "FMX_PAIR_FINISH"):
        if os.environ.get(opt):
            ix.set_option(opt[4:].lower(), int(os.environ[opt]))
    info = ix.info()
    log(f"[bench] index built+uploaded in {time.time() - t0:.1f}s: {info.n_suffixes} suffixes, "
        f"{info.device_bytes / 1e9:.2f} GB in HBM")

    params = fmx.Params(**pkw)
    cap = params.nmatch or 16

    class Workload:
        """One batch: ids and packed spellings in pinned host memory (+ a device copy of the ids for the device-resident legs)."""

        def __init__(self, q_off, q_tok, want_text):
            self.nq = len(q_off) - 1
            self.q_off = fmx.pinned_copy(q_off.astype(np.uint32))
            self.q_tok = fmx.pinned_copy(q_tok.astype(np.uint32))
            self.total_tokens = int(q_off[-1])
            self.max_len = int(np.diff(q_off).max()) if self.nq else 0
            self.d_off = self.d_tok = None
            self.text = None
            if want_text:
                need = int(L.fms_pack_text(self.q_tok.ctypes.data, len(self.q_tok), None, 0))
                self.text = synth.pack_text(self.q_tok, out=fmx.pinned_empty(max(need, 1), np.uint8))
            self.tok_ptrs = None

        def to_device(self):
            if self.d_off is None:
                self.d_off = torch.from_numpy(self.q_off.astype(np.int32)).to(dev)
                self.d_tok = torch.from_numpy(self.q_tok.astype(np.int32)).to(dev)

    t0 = time.time()
    main_set = [Workload(o, t, True) for k, (o, t) in enumerate(raw_main)]
    other_set = [Workload(o, t, True) for (o, t) in raw_other]
    del raw_main, raw_other
    nq_max = max([w.nq for w in main_set + other_set] + [1])
    # one result slot per call in flight, shared by the batches (step i writes slot i % T)
    out_slots = [(fmx.pinned_empty((nq_max, cap), fmx.MATCH_DTYPE), fmx.pinned_empty(nq_max, np.uint32), fmx.pinned_empty(nq_max, np.uint32))
                 for _ in range(max(T_host, T_dev, T_text))]
    log(f"[bench] batches pinned + packed as text in {time.time() - t0:.1f}s ({main_set[0].text.nbytes / 1e6:.0f} MB of spellings, "
        f"{main_set[0].q_tok.nbytes / 1e6:.0f} MB of ids per batch)")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the host threads that keep calls in flight live for the whole run: a new thread pays for its HIP thread state on its
    # first call (milliseconds), which must not land inside the timed region
    pool = cf.ThreadPoolExecutor(max(T_dev, 1))

    call_trace = []

    def run_host_steps(wls, k_steps, kind, gather=None):
        """K synchronous host-buffer calls, T_host of them in flight: step i (batch i % len(wls)) is submitted when step i - T has
        returned, so the copies at both ends of one call overlap the kernels of the others.  kind: "text" = fmx_match_batch_text
        (packed spellings in, ids looked up on the device: THE METRIC), "ids" = fmx_match_batch, "strings" = fmx_vocab_lookup_mt on
        host threads, then fmx_match_batch.  N > 1: each step's records also stay in HBM (the _keep entry points) and are gathered
        to rank 0 over RCCL; the gathers are launched by this thread in step order, so every rank issues its collectives in the
        same order."""
        T = T_host
        stats = []
        nb = len(wls)

        def one(i, keep):
            if os.environ.get("FMX_BENCH_TRACE_CALLS"):   # design-time: when each synchronous call ran, and on which thread
                import threading
                a = time.perf_counter()
                r = one_(i, keep)
                call_trace.append((kind, i, threading.get_ident() % 1000, a, time.perf_counter(), r.ms_total))
                return r
            return one_(i, keep)

        def one_(i, keep):
            wl = wls[i % nb]
            if kind == "text":
                return ix.match_batch_text(params, wl.q_off, wl.text, cap=cap, want_stats=True, out=out_slots[i % T], keep=keep)[3]
            if kind == "strings":
                a = time.perf_counter()
                ix.vocab_lookup_ptrs(wl.tok_ptrs, str_ids[i % T], str_threads)
                t_lookup[0] += time.perf_counter() - a
                return ix.match_batch(params, wl.q_off, str_ids[i % T][:len(wl.q_tok)], cap=cap, want_stats=True, out=out_slots[i % T], keep=keep)[3]
            return ix.match_batch(params, wl.q_off, wl.q_tok, cap=cap, want_stats=True, out=out_slots[i % T], keep=keep)[3]

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

    def settle(fn, n_round):
        """Set-up, not warm-up: every call in flight works in its own context, and a context sizes its workspace on its first
        batches (allocation, a re-run when a batch needs more than any before it).  Rounds of n_round untimed steps until two
        rounds in a row take no more than 1.25 x the fastest so far (at most eight; N > 1: four rounds, the same on every rank)."""
        if world > 1:
            for _ in range(4):
                fn(n_round)
            return
        best, ok = None, 0
        for _ in range(8):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn(n_round)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            ok = ok + 1 if (best is not None and dt <= 1.25 * best) else 0
            best = dt if best is None else min(best, dt)
            if ok >= 2:
                break

    def timed(fn, k_steps):
        # Python's cyclic garbage collector stays out of the timed regions (like timeit does it): a generation-2 collection of this
        # process takes ~37 ms — every thread of the harness stops, the library's calls do not — and landed inside one leg or
        # another in most runs: the "stalled" side legs of rounds 2-5 (found with FMX_BENCH_TRACE_CALLS, round 5)
        gc.collect()
        gc.disable()
        try:
            sync()
            t1 = time.perf_counter()
            res = fn(k_steps)
            sync()
            el = time.perf_counter() - t1
        finally:
            gc.enable()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    dev_slots = [(torch.zeros((nq_max, cap, 4), dtype=torch.int32, device=dev), torch.zeros((nq_max,), dtype=torch.int32, device=dev))
                 for _ in range(T_dev)]
    torch.cuda.synchronize()

    def run_device_pipeline(wls, k_steps, gather=None):
        """Side leg `device_resident`: K calls of the device entry point — query offsets / ids already in HBM when the timed region
        starts, records written to HBM — with T_dev calls in flight: ONE host thread enqueues them (fmx_match_batch_device_begin
        returns once a call's launches are queued, _end waits for it), every call works in its own context.  No per-stage events,
        no statistics.  Step i (batch i % len(wls)) is enqueued when step i - T has been waited for."""
        for wl in wls:
            wl.to_device()
        T = T_dev
        nb = len(wls)

        def begin(i):
            wl = wls[i % nb]
            if gather is not None:   # N > 1: the records are written where the RCCL gather of this step picks them up
                d_out, d_cnt = gather.buffers(i, host_sync=True)  # the slot's previous gather is over
            else:
                d_out, d_cnt = dev_slots[i % T]
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

    def run_text_pipeline(wls, k_steps, gather=None):
        """THE METRIC.  The text batches driven by ONE host thread: fmx_match_batch_text_begin enqueues a step's copies in, its id
        lookup and its passes and returns, _end waits and brings the records home; T_text steps in flight, each in its own context
        (step i, batch i % len(wls), begins when step i - T has ended).  N > 1: each step's records also stay in HBM and are gathered
        to rank 0 over RCCL; the gathers are launched in step order, so every rank issues its collectives in the same order."""
        T = T_text
        nb = len(wls)

        def begin(i):
            wl = wls[i % nb]
            keep = None
            if gather is not None:
                d_out, d_cnt = gather.buffers(i, host_sync=True)
                keep = (d_out.data_ptr(), d_cnt.data_ptr())
            return ix.text_begin(params, wl.q_off, wl.text, out_slots[i % T], cap, keep=keep)

        pending = {i: begin(i) for i in range(min(T, k_steps))}
        for i in range(k_steps):
            ix.text_end(pending.pop(i))
            if gather is not None:
                gather.launch(i)
            if i + T < k_steps:
                pending[i + T] = begin(i + T)
        if gather is not None:
            gather.wait_all()

    st_out = torch.zeros((nq_max, cap, 4), dtype=torch.int32, device=dev)
    st_cnt = torch.zeros((nq_max,), dtype=torch.int32, device=dev)

    def run_device_steps(wls, k_steps, first=0):
        """One call at a time with per-stage HIP events (stage times for the rooflines); step i runs batch (first + i) % len(wls)."""
        for wl in wls:
            wl.to_device()
        sts = []
        for i in range(k_steps):
            wl = wls[(first + i) % len(wls)]
            sts.append(ix.match_batch_device(params, wl.nq, wl.d_off.data_ptr(), wl.d_tok.data_ptr(), wl.total_tokens,
                                             st_out.data_ptr(), st_cnt.data_ptr(), 0, cap, stream=0, max_len=wl.max_len))
        return sts

    if args.child and os.environ.get("FMX_CHILD_MODE") == "pipeline":  # design-time: trace of the overlapped device-resident calls
        run_device_pipeline(main_set, 2 * T_dev + args.warmup)
        torch.cuda.synchronize()
        run_device_pipeline(main_set, args.steps)
        torch.cuda.synchronize()
        return
    if args.child and os.environ.get("FMX_CHILD_MODE") == "text":      # design-time: trace of the timed region of `value`
        run_text_pipeline(main_set[:T_text + 1], 2 * T_text + args.warmup)
        torch.cuda.synchronize()
        run_text_pipeline(main_set[:T_text + 1], args.steps)
        torch.cuda.synchronize()
        return
    if args.child:  # profiling child of a --pmc pass: device-resident steps one at a time, nothing else
        run_device_steps(main_set, args.warmup)
        torch.cuda.synchronize()
        run_device_steps(main_set, args.steps)
        torch.cuda.synchronize()
        return

    def make_gather(wls):
        if not (world > 1 or args.force_gather):
            return None
        return shard.RecordGather(wls[0].nq, cap, dev, dst=0, depth=T_dev + 1, coll_device=cdev)

    # what a leg's last timed step left behind, sampled for the parity check: {leg: (batch index, query indices, records, counts)}
    samples = {}
    # N > 1, value leg: (rank, that rank's block as CSR, query indices inside the block, gathered records, gathered counts)
    gathered_samples = []

    def sample_idx(nq):
        nv = min(args.verify, nq)
        return (np.arange(nv, dtype=np.int64) * nq // max(nv, 1)) if nv else np.zeros(0, np.int64)

    def keep_host_sample(leg, k_steps, wls):
        if rank != 0 or not args.verify:
            return
        i = k_steps - 1
        b = i % len(wls)
        idx = sample_idx(wls[b].nq)
        o, c, _ = out_slots[i % T_host]
        samples[leg] = (b, idx, o[idx].copy(), c[idx].copy())

    def measure_host(wls, kind, k_steps, leg):
        gather = make_gather(wls)
        settle(lambda k: run_host_steps(wls, k, kind, gather), 2 * T_host)
        run_host_steps(wls, args.warmup, kind, gather)
        el, sts = timed(lambda k: run_host_steps(wls, k, kind, gather), k_steps)
        if leg:
            keep_host_sample(leg, k_steps, wls)
        return el, sts

    def measure_device(wls, leg):
        gather = make_gather(wls)
        settle(lambda k: run_device_pipeline(wls, k, gather), 2 * T_dev)
        run_device_pipeline(wls, args.warmup, gather)
        el, _ = timed(lambda k: run_device_pipeline(wls, k, gather), args.steps)
        if gather is not None:  # the parity check reads the plain slots
            run_device_pipeline(wls, args.steps)
        if leg and rank == 0 and args.verify:
            i = args.steps - 1
            b = i % len(wls)
            idx = sample_idx(wls[b].nq)
            d_out, d_cnt = dev_slots[i % T_dev]
            tidx = torch.from_numpy(idx).to(dev)
            samples[leg] = (b, idx, d_out[tidx].cpu().numpy().view(fmx.MATCH_DTYPE).reshape(len(idx), cap), d_cnt[tidx].cpu().numpy().astype(np.uint32))
        return el

    def all_ranks(n):
        if world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        return int(t.item())

    host_set = main_set[:NB_host]
    text_set = main_set[:min(len(main_set), T_text + 1)]
    # ---- THE METRIC: SURVEY §8(d)'s bracket.  Packed spellings in pinned host memory -> H2D -> k_vocab_lookup -> pipeline -> D2H ----
    def measure_text(wls, leg):
        gather = make_gather(wls)
        # one host thread pipelining _begin / _end may keep more host-buffer calls active than the library admits by default (4:
        # the bound that keeps sixteen synchronous callers off the scheduler cliff); the option goes back after the leg
        ix.set_option("max_host_active", max(4, T_text))
        try:
            return measure_text_admitted(wls, leg, gather)
        finally:
            ix.set_option("max_host_active", 4)

    def measure_text_admitted(wls, leg, gather):
        settle(lambda k: run_text_pipeline(wls, k, gather), 2 * T_text)
        run_text_pipeline(wls, args.warmup, gather)
        el, _ = timed(lambda k: run_text_pipeline(wls, k, gather), args.steps)
        if leg and rank == 0 and args.verify:
            i = args.steps - 1
            b_ = i % len(wls)
            idx = sample_idx(wls[b_].nq)
            o_, c_, _t = out_slots[i % T_text]
            samples[leg] = (b_, idx, o_[idx].copy(), c_[idx].copy())
            if world > 1 and gather is not None and all_blocks is not None:
                # what the gather of the LAST timed step left on rank 0, in global query order: a spread sample of EVERY rank's block
                assert b_ == b_last_text
                g_out, g_cnt = gather.result(i)
                g_out = g_out.cpu().numpy().view(fmx.MATCH_DTYPE).reshape(-1, cap)
                g_cnt = g_cnt.cpu().numpy().astype(np.uint32)
                start = 0
                per = max(args.verify // world, 1)
                for r in range(world):
                    off_r, tok_r = all_blocks[r]
                    n_r = len(off_r) - 1
                    assert n_r == gather.sizes[r], (r, n_r, gather.sizes)
                    nv = min(per, n_r)
                    idx_r = np.arange(nv, dtype=np.int64) * n_r // max(nv, 1)
                    gathered_samples.append((r, off_r, tok_r, idx_r, g_out[start + idx_r].copy(), g_cnt[start + idx_r].copy()))
                    start += n_r
        return el

    elapsed = measure_text(text_set, "value")
    nq_step_all = all_ranks(text_set[0].nq)
    value = nq_step_all * args.steps / elapsed
    # ---- side legs ----
    # the same bracket from T_host synchronous callers (fmx_match_batch_text on as many host threads: the reference's calling pattern)
    # (the id leg BEFORE the synchronous text callers: measured after them it ran at a third of its speed in three runs out of four —
    # its calls one at a time instead of four at once — and never when it came first, tools/dbg/ids_callers.py; not understood)
    elapsed_ids, ids_stats = measure_host(host_set, "ids", args.steps, "pcie_inclusive_ids")
    value_ids = nq_step_all * args.steps / elapsed_ids
    el_sync, text_stats = measure_host(host_set, "text", args.steps, "text_sync_callers")
    if call_trace:
        last = call_trace[-args.steps:]
        t0_ = min(x[3] for x in last)
        for x in last:
            log(f"[trace] {x[0]} call {x[1]:3d} thread {x[2]:3d}: {1e3 * (x[3] - t0_):8.2f} -> {1e3 * (x[4] - t0_):8.2f} ms (library: {x[5]:.2f} ms)")
    text_sync = {"value": round(nq_step_all * args.steps / el_sync, 1), "ms_per_step": round(el_sync / args.steps * 1e3, 3), "calls_in_flight": T_host,
                 "note": "the value leg's bracket with synchronous calls: fmx_match_batch_text from as many host threads"}
    el2 = measure_text(other_set[:min(len(other_set), T_text + 1)], None) if world > 1 else None

    from_strings = None
    str_ids, str_threads, t_lookup = [], 1, [0.0]
    if world == 1 and not args.no_text_leg:
        # SURVEY §8(d)'s bracket with the id lookup on HOST threads — VocabIndexer::getIndex per pattern token (src/fuzzy_match.cc:
        # 468-470) as fmx_vocab_lookup_mt over one char pointer per token — then the host-buffer call.  Two batches rotate (a batch
        # of pointers is 200 MB).
        import oracle_lib as _ol
        cores = _ol.effective_cpus()
        str_threads = max(1, cores // T_host)
        str_set = host_set[:2]
        for wl in str_set:
            wl.tok_ptrs = np.ascontiguousarray(word_ptrs[wl.q_tok])          # one char pointer per pattern token of the batch
        str_ids = [fmx.pinned_empty(max(len(w.q_tok) for w in str_set), np.uint32) for _ in range(T_host)]
        run_host_steps(str_set, T_host, "strings")
        assert np.array_equal(str_ids[(T_host - 1) % T_host][:len(str_set[(T_host - 1) % 2].q_tok)], str_set[(T_host - 1) % 2].q_tok), "vocabulary lookup does not reproduce the ids"
        t_lookup[0] = 0.0
        k = max(6, args.steps // 2)
        el, _ = timed(lambda kk: run_host_steps(str_set, kk, "strings"), k)
        from_strings = {"value": round(str_set[0].nq * k / el, 1), "ms_per_step": round(el / k * 1e3, 3), "steps": k,
                        "lookup_ms_per_step": round(t_lookup[0] / k * 1e3, 3), "lookup_threads_per_call": str_threads, "calls_in_flight": T_host,
                        "tokens_per_step": int(len(str_set[0].q_tok)),
                        "note": "fmx_vocab_lookup_mt (token strings -> ids on host threads, in the timed region) + fmx_match_batch (pinned host buffers)"}
        for wl in str_set:
            wl.tok_ptrs = None

    # ---- the drop-in CLI end to end on the same workload: stdin -> stdout through tools/fmx-cli --tokenised ----
    cli_leg = cli_raw_leg = None
    if world == 1 and not args.no_text_leg:
        try:
            cli_leg = cli_end_to_end(ix, host_set[0], pkw, log)
        except Exception as e:  # noqa: BLE001  (a side leg: the line is still worth printing)
            log(f"[bench] cli_end_to_end leg failed: {e!r}")
        try:   # the same queries as RAW text (no --tokenised): the reference CLI's only mode
            cli_raw_leg = cli_end_to_end(ix, host_set[0], pkw, log, tokenised=False)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cli_raw_text leg failed: {e!r}")

    # ---- the reference CLI's DEFAULT kind of index (built from cased text under -p nbr,tag,cas: real tokens differ from the normalised
    # tokens, every candidate pays the real-token term of _edit_distance): tools/default_index_leg.py ----
    default_leg = None
    if world == 1 and not args.no_default_index and not args.no_text_leg:
        try:
            import default_index_leg
            default_leg = default_index_leg.run(fmx, synth, torch, dev, c_off, c_tok, [(w.q_off, w.q_tok) for w in text_set], pkw, cap, T_text, args.steps,
                                                args.warmup, args.verify, log, id_only_value=value)
        except Exception as e:  # noqa: BLE001  (a side leg: its parity mismatches are in its own dict, like the other legs')
            log(f"[bench] cli_default_index leg failed: {e!r}")

    # ---- one call at a time from HBM (device entry point with statistics): stage times for the rooflines ----
    run_device_steps(main_set, 1)
    el_dev, dev_stats = timed(lambda k: run_device_steps(main_set, k), args.steps)
    if rank == 0 and args.verify:
        b = (args.steps - 1) % len(main_set)
        idx = sample_idx(main_set[b].nq)
        tidx = torch.from_numpy(idx).to(dev)
        samples["staged"] = (b, idx, st_out[tidx].cpu().numpy().view(fmx.MATCH_DTYPE).reshape(len(idx), cap), st_cnt[tidx].cpu().numpy().astype(np.uint32))
    names = fmx.stage_names()
    stage_ms = np.zeros(16)
    for st in dev_stats:
        stage_ms += np.array(list(st.ms_kernel))
    per_stage = {n: float(stage_ms[i] / max(args.steps, 1)) for i, n in enumerate(names)}
    K = max(len(dev_stats), 1)

    def mean_ctr(name):   # kernel counters, averaged over the staged steps (the batches rotate)
        return sum(float(getattr(s, name)) for s in dev_stats) / K

    max_kept = max(int(s.max_kept) for s in dev_stats)
    # bytes / lines the probe's own loads request: the counting variant of the kernel, one untimed step per distinct batch
    ix.set_option("probe_count", 1)
    csts = run_device_steps(main_set, len(main_set))
    ix.set_option("probe_count", 0)
    torch.cuda.synchronize()
    # ---- the six-call device-resident leg LAST: whatever is measured after it (its contexts' queues still recently active) stalls
    # now and then (DESIGN §5 "Stalls") ----
    elapsed_dev = measure_device(main_set, "device_resident")
    value_dev = nq_step_all * args.steps / elapsed_dev
    scaling_other = None
    if world > 1:
        el2d = measure_device(other_set, None)
        n2 = all_ranks(other_set[0].nq)
        scaling_other = {"scaling": other_scaling, "value": round(n2 * args.steps / el2, 1),
                         "ms_per_step": round(el2 / args.steps * 1e3, 3), "queries_per_step_all_ranks": n2,
                         "device_resident_value": round(n2 * args.steps / el2d, 1)}
    probe_bytes = sum(float(s.probe_bytes) for s in csts) / len(csts)
    probe_lines = sum(float(s.probe_lines) for s in csts) / len(csts)
    tot_tokens = sum(w.total_tokens for w in main_set) / len(main_set)

    # stage "expand_a" = k_expand_a and k_expand_h side by side (main stream: until both are done); "expand_l" = the second
    # scan + k_expand_b<true> with k_expand_b<false> beside it, "expand_wait" = what is left of the latter afterwards
    probe_ms, exa_ms = per_stage.get("probe", 0.0), per_stage.get("expand_a", 0.0)
    # byte models of THIS algorithm, per launch, from the kernels' own counters:
    #   k_probe     bytes its loads request (n-gram, qva pair, table buckets, key tails, ...)
    #   expand_a    k_expand_a: 8 B per hit it walks (short ranges of the queries that are not light); k_expand_h: 8 B per
    #               plane word + 8 B per hit re-tested exactly; both: 8 B per hit queued for k_expand_b<false>; 8 B per
    #               (query, start) range descriptor of the queries they visit (upper bound: all)
    b_probe = probe_bytes
    n_hits, n_heavy_hits, n_light_hits = mean_ctr("n_hits"), mean_ctr("n_heavy_hits"), mean_ctr("n_light_hits")
    n_phase_b, n_kept, n_cand, n_rerank = mean_ctr("n_phase_b"), mean_ctr("n_kept"), mean_ctr("n_candidates"), mean_ctr("n_rerank")
    walked_a = n_hits - n_heavy_hits - n_light_hits
    queued = n_phase_b - n_light_hits
    b_exa = 8.0 * walked_a + 8.0 * mean_ctr("plane_words") + 8.0 * mean_ctr("n_heavy_pass") + 8.0 * queued + 8.0 * tot_tokens
    #   expand_b    k_expand_b<true> (light queries): 8 B range descriptor per start position + 16 B record per hit;
    #               k_expand_b<false>: 8 B queue entry + 16 B record per queued hit; both: the pattern once per query, and for
    #               every hit kept the sentence's tokens (mean length of the corpus) and a 16 B segment entry; the tokens of
    #               hits that fail the dominance / coverage tests are not counted (a lower bound of what the kernels read)
    exb_ms = per_stage.get("expand_l", 0.0) + per_stage.get("expand_wait", 0.0)
    b_exb = (16.0 * n_light_hits + 24.0 * queued + 12.0 * tot_tokens + n_kept * (4.0 * (mean_len + 1) + 16.0))
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
        log(f"[bench] value (packed spellings in pinned host memory -> records in pinned host memory, {T_text} text batches in flight from one host thread): {value / 1e6:.2f} M patterns/s, "
            f"{elapsed / args.steps * 1e3:.3f} ms/step; device-resident ({T_dev} in flight): {value_dev / 1e6:.2f} M, {elapsed_dev / args.steps * 1e3:.3f} ms/step; "
            f"from ids in pinned host memory: {value_ids / 1e6:.2f} M, {elapsed_ids / args.steps * 1e3:.3f} ms/step (pinned={ids_stats[-1].pinned}); "
            f"one call at a time with stage events: {main_set[0].nq * args.steps / el_dev / 1e6:.2f} M patterns/s")
        log(f"[bench] the same from {T_host} synchronous callers: {text_sync['value'] / 1e6:.2f} M patterns/s, {text_sync['ms_per_step']:.3f} ms/step")
        log(f"[bench] synchronous text calls: retries {sum(int(s.n_retries) for s in text_stats)}, ms per call min/median/max "
            f"{min(s.ms_total for s in text_stats):.2f}/{float(np.median([s.ms_total for s in text_stats])):.2f}/{max(s.ms_total for s in text_stats):.2f}")
        log(f"[bench] synchronous id calls: retries {sum(int(s.n_retries) for s in ids_stats)}, ms per call min/median/max "
            f"{min(s.ms_total for s in ids_stats):.2f}/{float(np.median([s.ms_total for s in ids_stats])):.2f}/{max(s.ms_total for s in ids_stats):.2f}; "
            f"host ms in the copies' enqueue, median: in {float(np.median([s.ms_h2d for s in ids_stats])):.3f} out {float(np.median([s.ms_d2h for s in ids_stats])):.3f}")
        log(f"[bench] kernel ms/step by stage: { {k: round(v, 3) for k, v in per_stage.items()} }")
        log(f"[bench] mean step: hits {n_hits:.0f} (bit planes {n_heavy_hits:.0f}, passed {mean_ctr('n_heavy_pass'):.0f}) phaseB {n_phase_b:.0f} "
            f"(light queries {n_light_hits:.0f}) kept {n_kept:.0f} candidates {n_cand:.0f} retries {sum(int(s.n_retries) for s in dev_stats)} "
            f"big {mean_ctr('n_big'):.1f} max_kept {max_kept}")

        # ---- HBM traffic of the kernels, measured now: rocprofv3 --pmc child passes (separate passes) ----
        traffic = {}
        if not args.no_pmc:
            # N > 1: the child passes run here on rank 0 only (one process, this rank's GPU, this rank's share of the step), the other
            # ranks wait at the barrier that ends the run — `roofline.traffic` is measured at every N (north_star: "rocprof HBM GB/s
            # vs peak reported at 1/2/4/8 GPUs")
            child = ["--config", args.config, "--steps", "2", "--warmup", "1", "--scaling", "weak", "--threads", str(args.threads),
                     "--host-threads", str(args.host_threads)]
            if args.sentences:
                child += ["--sentences", str(args.sentences)]
            if world > 1 or args.queries:
                child += ["--queries", str(main_set[0].nq)]
            cal = json.load(open(CALIBRATION)) if os.path.exists(CALIBRATION) else {}
            t1 = time.time()
            fetch = pmc_pass("FETCH_SIZE", child, log)
            write = pmc_pass("WRITE_SIZE", child, log) if fetch else None
            log(f"[bench] pmc child passes: {time.time() - t1:.0f}s")
            if fetch and write:
                rounds = int(pkw.get("nmatch", 1)) if pkw.get("contrast", 0.0) > 0.0 else 1
                for kern, pattern, per_step in (("fmx::k_probe", "random", 1), ("fmx::k_expand_a", "stream16", 1), ("fmx::k_expand_h", "stream8", 1),
                                                ("fmx::k_expand_b", "random", 1), ("fmx::k_dp", "random", 1), ("fmx::k_finish", "random", 1),
                                                ("fmx::k_camap", "random", 1), ("fmx::k_cround", "random", rounds), ("fmx::k_crerank_g", "random", 1)):
                    keys = [k for k in fetch if k.replace("void ", "").startswith(kern)]  # template instantiations add up
                    if not keys:
                        continue
                    # the child's two timed steps (the last dispatches: earlier ones include re-run first passes): bytes per step.
                    # The light kernel (k_expand_b<true>) runs as two launches per pass of >= 65 536 queries since round 6.
                    def last_two_steps(name, series):
                        n_ps = per_step * (2 if ("k_expand_b<true>" in name and main_set[0].nq >= 65536) else 1)
                        return np.sum([v for _, v in series[-2 * n_ps:]]) / 2.0
                    f_kb = sum(last_two_steps(k, fetch[k]) for k in keys)
                    w_kb = sum(last_two_steps(k, write[k]) for k in keys if k in write)
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

        r_probe = roof("k_probe+k_probe_tail", b_probe, probe_ms, ["k_probe"],
                       "bytes requested by the two kernels' own loads, counted by their COUNT variants on the same batches: seed n-gram, "
                       "_quickVocabAccess pair, n-gram table buckets (head: a lane per start position), 7-gram entries and 32 B keys of the "
                       "bisection (tail: a lane per start position the head could not decide), the 8 B list entry in between")
        r_exa = roof("k_expand_h+k_expand_a", b_exa, exa_ms, ["k_expand_a", "k_expand_h"],
                     "phase A of the queries with long ranges or many hits (two kernels side by side): 8 B per bit-plane word, "
                     "8 B per plane survivor re-tested, 8 B per hit walked one by one, 8 B per hit queued, 8 B per range "
                     "descriptor (kernel counters)")
        if r_exa["traffic"]:
            # These two kernels REQUEST more bytes than reach HBM (neighbouring work items share plane rows in L2): a model above the
            # counted traffic is not an HBM fraction.  One number: the counted traffic; the requested bytes stay as a note.
            r_exa.update({"achieved": r_exa["traffic_gbps"], "frac": r_exa["traffic_frac"], "requested_bytes_per_launch": r_exa["algorithmic_bytes_per_launch"],
                          "algorithmic_bytes_per_launch": r_exa["traffic"],
                          "model": "HBM bytes counted by rocprofv3 (FETCH_SIZE / WRITE_SIZE passes of this run); the kernels request more — "
                                   "requested_bytes_per_launch: " + r_exa["model"]})
        r_probe["lines_bytes_per_launch"] = int(probe_lines) * 128  # 128 B lines its loads touch x 128: what random reads cost the memory system
        r_exb = roof("k_expand_b<true>+k_expand_b<false>", b_exb, exb_ms, ["k_expand_b"],
                     "hit walk of the light queries + phase B of the others (side by side): 8 B range descriptor per start, 16 B "
                     "record per light hit, 8 B queue entry + 16 B record per queued hit, 4 B per pattern token, and per kept "
                     "hit its sentence (4 B x (mean length + 1)) and a 16 B segment entry; reads for hits that fail the "
                     "dominance / length tests are not counted (lower bound)")
        # edit distances: per candidate its sentence (length word + tokens), the 16 B candidate record read and written back, the
        # 8 B list entry; the pattern is shared by the candidates of a query (4 B per pattern token per query)
        dp_ms = per_stage.get("dp", 0.0)
        b_dp = n_cand * (4.0 * (mean_len + 1) + 16.0 + 16.0 + 8.0) + 4.0 * tot_tokens
        r_dp = roof("k_dp_tab+k_dp_reg<64>", b_dp, dp_ms, ["k_dp"],
                    "one lane per (query, candidate): 4 B x (sentence length + 1) of the sentence, 16 B candidate record in and out, 8 B "
                    "list entry, 4 B per pattern token per query (kernel counters; mean sentence length of the corpus)")
        roofs = [r_probe, r_exa, r_exb, r_dp]
        if pkw.get("contrast", 0.0) > 0.0:
            # the contrastive rerank: per sentence-pair distance both sentences and the per-match state (list entry 8, record 16,
            # keys 8 + 8, penalty sum 4 + 4, penalty 4, two sentence positions 8); per candidate the 16 B record the replay reads
            fin_ms = per_stage.get("finish", 0.0)
            b_fin = n_rerank * (8.0 * (mean_len + 1) + 64.0) + 16.0 * n_cand
            roofs.append(roof("finish stage: replay + k_crerank_g<8|16|32|64> (+ k_camap, nmatch x k_cround<64> for sentences of 33-64 tokens)", b_fin, fin_ms,
                              ["k_finish", "k_crerank_g", "k_camap", "k_cround"],
                              f"replay + contrastive rerank; {int(n_rerank)} sentence-pair edit distances per step (kernel counter) x (two sentences "
                              "of 4 B x (mean length + 1) + 64 B of per-match state) + 16 B per candidate; the group kernels keep the sentences in "
                              "registers over the rounds, so this is an upper bound of what they request; the stage is bound by VALU issue (a third "
                              "of the group kernels' lanes have work), not by HBM — priced here because the line's schema has no issue roofline"))
        # the dominant kernel group = the longest stage, whatever it is (C5: the finish stage)
        roofs.sort(key=lambda r: -r["kernel_ms"])
        dominant, others = roofs[0], roofs[1:]

        # ---- parity check, reference byte model, CPU baseline (the oracle: test infrastructure, outside the timed region) ----
        import oracle_lib
        t1 = time.time()
        o = oracle_lib.OracleIndex()
        o.set_arrays(ix.array("suf_sid"), ix.array("suf_pos"), ix.array("buffer"), ix.array("sent_pos"),
                     ix.array("qva"), int(ix.info().vocab_size))   # (the oracle counts the sentence frequencies itself)
        log(f"[bench] oracle index ready in {time.time() - t1:.1f}s")
        oprm = oracle_lib.Params(**pkw)
        parity = ref_model = cpu_baseline = None
        if samples:
            bad = {}
            want = {}
            for leg, (b, idx, recs, cnts) in samples.items():
                wl = main_set[b]   # (every leg's batches are a prefix of main_set)
                key = (id(wl), len(idx))
                if key not in want:
                    s_off, s_tok = take_queries(wl.q_off, wl.q_tok, idx)
                    want[key] = o.match_batch(oprm, s_off, s_tok, cap, nthreads=oracle_lib.effective_cpus())
                oo, oc = want[key]
                bad[leg] = sum(int(oc[q] != cnts[q] or oo[q, :oc[q]].tobytes() != recs[q, :cnts[q]].tobytes()) for q in range(len(idx)))
            per_rank_block = None
            if gathered_samples:
                per_rank_block = []
                for r, off_r, tok_r, idx_r, recs, cnts in gathered_samples:
                    s_off, s_tok = take_queries(off_r, tok_r, idx_r)
                    oo, oc = o.match_batch(oprm, s_off, s_tok, cap, nthreads=oracle_lib.effective_cpus())
                    nbad = sum(int(oc[q] != cnts[q] or oo[q, :oc[q]].tobytes() != recs[q, :cnts[q]].tobytes()) for q in range(len(idx_r)))
                    per_rank_block.append({"rank": r, "queries_in_block": int(len(off_r) - 1), "checked": int(len(idx_r)), "mismatches": int(nbad)})
            parity = {"checked": int(min(args.verify, host_set[0].nq)),
                      "mismatches": int(sum(bad.values()) + sum(x["mismatches"] for x in (per_rank_block or []))), "per_leg": bad,
                      "per_rank_block": per_rank_block,
                      "paths": "what the LAST timed step of each leg left behind (the value leg, the device-resident pipeline, the id batches, the "
                               "staged calls), every (nq / checked)-th query of that step's batch, vs the oracle; N > 1: per_rank_block = what "
                               "the RCCL gather of the value leg's last timed step assembled on rank 0, a spread sample of every rank's block "
                               "(global query order, cli/src/FuzzyMatch-cli.cc:151-162) vs the oracle"}
        if parity is not None and default_leg and default_leg.get("parity"):
            # the default-index leg checked its three paths against the oracle WITH real tokens itself (tools/default_index_leg.py)
            parity["per_leg"]["cli_default_index"] = int(default_leg["parity"].get("mismatches", 0))
            parity["mismatches"] += parity["per_leg"]["cli_default_index"]
        q_off, q_tok, nq = main_set[0].q_off, main_set[0].q_tok, main_set[0].nq
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
            # the same with ONE thread ("with N = all host cores ... and with N = 1"), a few seconds of it
            # (round 6: every (ns2 / ns1)-th query of the SAME ns2 queries, after the all-threads run has walked the index)
            ns1 = int(min(ns2, max(500, rate / max(cores, 1) * min(args.cpu_seconds, 8.0))))
            i1 = (np.arange(ns1, dtype=np.int64) * ns2 // max(ns1, 1))
            o1, t1q = take_queries(q_off, q_tok, i1)
            t1 = time.perf_counter()
            o.match_batch(oprm, o1, t1q, cap, nthreads=1)
            dt1 = time.perf_counter() - t1
            cpu_model = ""
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            cpu_baseline = {"value": round(ns2 / dt, 1), "unit": "patterns/s", "cores": cores, "kind": "port",
                            "value_n1": round(ns1 / dt1, 1), "cpu_model": cpu_model, "hardware_threads": os.cpu_count(),
                            "cpu_quota": f"{cores} of {os.cpu_count()} hardware threads usable (min(affinity, cgroup cpu.max)); the oracle scales "
                                         "to the quota and not beyond",
                            "sample": f"first {ns2} queries of batch 0 of the same {args.config} workload, oracle/fm_oracle.c "
                                      f"fmo_match_batch with {cores} threads (= usable CPUs: min(affinity, cgroup cpu.max) of "
                                      f"{os.cpu_count()} hardware threads), ids in (no lookup), {dt:.1f}s; value_n1: {ns1} of the same {ns2} queries (every "
                                      f"{ns2 / max(ns1, 1):.1f}-th) with one thread, afterwards, {dt1:.1f}s",
                            "sample_queries": int(ns2), "sample_queries_n1": int(ns1)}
        o.close()

        text_bytes = int(text_set[0].text.nbytes)
        line = {
            "metric": "query patterns/sec @ fuzzy=0.7, 5M-sentence .fmi; 1/2/4/8 GPU + CPU ref",
            "value": round(value, 1), "unit": "patterns/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n_sent}-sentence synthetic corpus (mean {mean_len} tok, near-duplicate families), "
                                   f"{nq_cfg} queries per {'GPU' if args.scaling == 'weak' else 'job'} per step, "
                                   + " ".join(f"{k}={v}" for k, v in pkw.items()),
                       "queries_per_step_all_ranks": nq_step_all, "sentences": n_sent, "index_bytes_hbm": int(info.device_bytes),
                       "timed_path": "SURVEY §8(d)'s bracket (cli/src/FuzzyMatch-cli.cc:450-467), PCIe inside: one text batch per step (fmx_match_batch_text_begin / _end) — "
                                     f"packed token spellings in pinned host memory ({text_bytes} B per step) -> H2D -> vocabulary id lookup on the device "
                                     f"(k_vocab_lookup) -> every kernel of the pipeline -> D2H of the top-N records into pinned host memory; {T_text} step(s) in flight, "
                                     f"enqueued and collected by one host thread, each in its own context; {len(text_set)} distinct query batches rotate through the steps"
                                     + (", + RCCL gather of every step's records to rank 0 (asynchronous, under the steps in flight)" if world > 1 else ""),
                       "distinct_batches": {"value_leg": len(text_set), "device_resident": len(main_set), "seeds": f"QUERY_SEED + k{' + 101 x rank' if args.scaling == 'weak' else ''}"},
                       "index_via_fmi": via_fmi,
                       "parallelism": f"queries sharded over {world} GPU(s), index replicated"},
            "roofline": dominant, "roofline_other": others, "reference_byte_model": ref_model,
            "device_resident": {"value": round(value_dev, 1), "ms_per_step": round(elapsed_dev / args.steps * 1e3, 3), "calls_in_flight": T_dev,
                                "note": "NOT the metric's bracket: fmx_match_batch_device_begin / _end, query offsets / ids resident in HBM when the timed "
                                        "region starts, records left in HBM (no lookup, no PCIe); one host thread enqueues, each call in its own context"},
            "pcie_inclusive_ids": {"value": round(value_ids, 1), "ms_per_step": round(elapsed_ids / args.steps * 1e3, 3), "calls_in_flight": T_host,
                                   "note": "fmx_match_batch: vocabulary ids in pinned host memory in, records out (H2D + kernels + D2H"
                                           + (" + the RCCL gather" if world > 1 else "") + "): a caller that already holds ids"},
            "text_sync_callers": text_sync, "from_token_strings": from_strings, "cli_end_to_end": cli_leg, "cli_raw_text": cli_raw_leg,
            "cli_default_index": default_leg,
            "staged": {"value": round(nq_step_all * args.steps / el_dev, 1),
                       "ms_per_step": round(el_dev / args.steps * 1e3, 3),
                       "stage_ms_per_step": {k: round(v, 4) for k, v in per_stage.items()},
                       "note": "device-resident, ONE call at a time, every pass synchronised and timed per stage with HIP events on "
                               "the launch stream (the kernel times the rooflines are priced with); batches rotate here too"},
            "scaling_other": scaling_other, "per_rank": rank_rates,
            "cpu_baseline": cpu_baseline, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cli_end_to_end(ix, wl, pkw, log, tokenised=True):
    """The drop-in CLI (tools/fmx-cli: the reference CLI's flags, output and stderr lines, cli/src/FuzzyMatch-cli.cc) on this run's
    own index and first query batch, the way the reference's CI runs it: `-i index.fmi -a match < queries > matches`.  The index goes
    through a .fmi file, the queries are a text file of blank-separated tokens.  Reported: patterns / second of the phase the
    reference brackets with TICK("Matching") .. TICK("Done") (cli:450-467: read, match, format, print), from the CLI's own STEP
    lines; the index load is outside, as it is there."""
    cli = os.path.join(ROOT, "tools", "fmx-cli")
    if not os.path.exists(cli):
        return None
    tmp = tempfile.mkdtemp(prefix="fmx_cli_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        fmi, qf, of = os.path.join(tmp, "index.fmi"), os.path.join(tmp, "queries.tok"), os.path.join(tmp, "matches.tsv")
        ix.save(fmi)
        lines = np.array(wl.text, dtype=np.uint8, copy=True)        # packed spellings -> one line of blank-separated tokens per pattern
        nul = np.flatnonzero(lines == 0)
        lines[nul] = 32
        lines[nul[wl.q_off[1:].astype(np.int64) - 1]] = 10
        lines.tofile(qf)
        argv = [cli] + (["--tokenised"] if tokenised else []) + ["-i", fmi, "-a", "match", "-f", str(pkw.get("fuzzy", 0.8)), "-n", str(pkw.get("nmatch", 5)),
                "--mr", str(pkw.get("min_subseq_ratio", 0.3)), "-I", str(pkw.get("idf_penalty", 0.0)), "--contrast", str(pkw.get("contrast", 0.0)),
                "--contrast-buffer", str(pkw.get("contrast_buffer", -1))]
        t0 = time.time()
        r = subprocess.run(argv, stdin=open(qf, "rb"), stdout=open(of, "wb"), stderr=subprocess.PIPE, text=True, timeout=600)
        wall = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-500:])
        steps = {}
        for ln in r.stderr.splitlines():
            f = ln.split("\t")
            if len(f) >= 6 and f[0] == "STEP":
                steps[f[1]] = float(f[5])
        nmatch = [ln for ln in r.stderr.splitlines() if ln.startswith("NMATCH")]
        phase = steps["Done"] - steps["Matching"]
        out_bytes = os.path.getsize(of)
        log(f"[bench] fmx-cli{' --tokenised' if tokenised else ' (raw text)'}: {wl.nq} patterns in {phase * 1e3:.0f} ms of matching phase = {wl.nq / phase / 1e6:.2f} M patterns/s "
            f"(process wall {wall:.1f}s with the index load); {nmatch[-1] if nmatch else ''}")
        setup = ("Before 'Matching' (like the index load, and like the reference sets its thread pool up — though the reference does that INSIDE "
                 "its bracket, cli/src/FuzzyMatch-cli.cc:139-147): the blocks' pinned buffers, the library's call contexts and four warm-up calls of "
                 "up to 65 536 patterns each (the memory's own sentences as id batches, and a block of dummy lines through the line entry point) that "
                 "size the workspaces; the worker threads are joined after 'Done'.")
        if tokenised:
            note = ("tools/fmx-cli --tokenised -i index.fmi -a match < queries > matches (--tokenised is a flag of THIS CLI: the reference has no "
                    "such mode): stdin -> fmx_match_batch_lines (the device cuts the lines into patterns and tokens and looks the ids up) -> formatted "
                    "records on stdout, in input order; time between the CLI's own 'Matching' and 'Done' STEP lines.  " + setup)
        else:
            note = ("tools/fmx-cli -i index.fmi -a match < queries > matches — RAW TEXT, the reference CLI's only mode (cli/src/FuzzyMatch-cli.cc:219-233 "
                    "-> src/fuzzy_match.cc:393-412): stdin -> fmx_match_batch_raw_lines (lines of ASCII letters / digits / punctuation / blanks are "
                    "tokenised, case-folded and looked up ON THE DEVICE; any other line by the host front-end) -> formatted records on stdout, in input "
                    "order; time between the CLI's own 'Matching' and 'Done' STEP lines.  The synthetic queries are lower-case ASCII words, so every "
                    "line takes the device path here.  " + setup)
        return {"value": round(wl.nq / phase, 1), "unit": "patterns/s", "patterns": int(wl.nq), "matching_phase_s": round(phase, 4),
                "process_wall_s": round(wall, 2), "input_bytes": int(lines.nbytes), "output_bytes": int(out_bytes), "nmatch_line": nmatch[-1] if nmatch else None,
                "note": note}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def through_fmi(fmx, ix, device, rank, log):
    """--via-fmi: the index as the metric names it — a .fmi file.  fmx_index_save writes the reference's on-disk format
    (src/fuzzy_matcher_binarization.cc:16-30), fmx_index_load reads it back (:32-51); every array of the loaded index is compared with
    the one that was saved.  Returns the loaded index (on the device) and the figures."""
    path = os.path.join(tempfile.gettempdir(), f"fmx_bench_{os.getpid()}_{rank}.fmi")
    try:
        t0 = time.time()
        ix.save(path)
        t_save = time.time() - t0
        size = os.path.getsize(path)
        t0 = time.time()
        ld = fmx.Index.load(path)
        t_load = time.time() - t0
        same = True
        for name in ("suf_sid", "suf_pos", "buffer", "sent_pos", "qva", "sfreq"):
            same = same and np.array_equal(ix.array(name), ld.array(name))
        assert same, "the index loaded from the .fmi differs from the one that was saved"
        a, b = ix.info(), ld.info()
        assert (a.n_sentences, a.n_suffixes, a.vocab_size, a.max_tokens_in_pattern, a.pt) == (b.n_sentences, b.n_suffixes, b.vocab_size, b.max_tokens_in_pattern, b.pt)
        ix.close()
        t0 = time.time()
        ld.finalize()
        ld.to_device(device)
        t_up = time.time() - t0
        log(f"[bench] .fmi: {size / 1e9:.2f} GB, saved in {t_save:.1f}s, loaded in {t_load:.1f}s, finalized + uploaded in {t_up:.1f}s; arrays equal")
        return {"index": ld, "file_bytes": int(size), "save_s": round(t_save, 2), "load_s": round(t_load, 2), "finalize_upload_s": round(t_up, 2),
                "arrays_equal": True}
    finally:
        if os.path.exists(path):
            os.unlink(path)


if __name__ == "__main__":
    main()
