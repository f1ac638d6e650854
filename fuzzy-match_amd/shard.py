"""Multi-GPU plumbing of the match() path: queries shard by rank, the index is replicated,
and the only collective is one gather of the fixed-size top-N records to rank 0.  Backend-agnostic
(`nccl` = RCCL over xGMI on the GPUs, `gloo` in the CPU tests); torch is plumbing only."""
import numpy as np


def shard_range(nq, rank, world):
    """Contiguous block of queries of `rank`: [g*nq/G, (g+1)*nq/G)."""
    return (nq * rank) // world, (nq * (rank + 1)) // world


def shard_batch(q_off, q_tok, rank, world):
    """CSR slice (offsets rebased to 0) of this rank's block."""
    lo, hi = shard_range(len(q_off) - 1, rank, world)
    off = np.asarray(q_off[lo:hi + 1], dtype=np.uint32)
    tok = np.asarray(q_tok[int(off[0]):int(off[-1])], dtype=np.uint32)
    return (off - off[0]).astype(np.uint32), tok, lo, hi


def gather_records(out, cnt, dst=0):
    """Gathers per-rank (nq_r, cap, 4) int32 record tensors and (nq_r,) counts to rank `dst`.
    Ranks may hold different numbers of queries; returns (records, counts) in global query order on
    `dst`, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return out, cnt
    rank = dist.get_rank()
    n = torch.tensor([out.shape[0]], dtype=torch.int64, device=out.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad_out = torch.zeros((mx,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    pad_cnt = torch.zeros((mx,), dtype=cnt.dtype, device=cnt.device)
    pad_out[:out.shape[0]] = out
    pad_cnt[:cnt.shape[0]] = cnt
    g_out = [torch.empty_like(pad_out) for _ in range(world)] if rank == dst else None
    g_cnt = [torch.empty_like(pad_cnt) for _ in range(world)] if rank == dst else None
    dist.gather(pad_out, g_out, dst=dst)
    dist.gather(pad_cnt, g_cnt, dst=dst)
    if rank != dst:
        return None, None
    return (torch.cat([g_out[r][:sizes[r]] for r in range(world)]),
            torch.cat([g_cnt[r][:sizes[r]] for r in range(world)]))


class RecordGather:
    """Persistent, double-buffered gather of the top-N records for a stream of equally shaped steps.

    Step i's kernels write into `buffers(i)`; `launch(i)` starts the gather of those buffers to rank `dst`
    asynchronously (RCCL runs it on its own stream over the point-to-point xGMI links), so it overlaps the
    compute of step i + 1, which uses the other slot.  A slot is only handed out again after its previous
    gather has completed (stream-ordered wait, no host sync).  Sizes are exchanged once, not per step."""

    def __init__(self, nq_local, cap, device, dst=0, depth=2, coll_device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst, self.nq, self.cap = dst, nq_local, cap
        # coll_device: where the collectives' tensors live when that is not `device` (gloo dry runs: host memory, synchronous)
        self.staged = coll_device is not None and coll_device != device
        if self.world > 1:
            n = torch.tensor([nq_local], dtype=torch.int64, device=coll_device if self.staged else device)
            sizes = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(sizes, n)
            self.sizes = [int(x.item()) for x in sizes]
        else:
            self.sizes = [nq_local]
        self.mx = max(self.sizes + [1])
        self.slots = []
        for _ in range(depth):
            slot = {"out": torch.zeros((self.mx, cap, 4), dtype=torch.int32, device=device),
                    "cnt": torch.zeros((self.mx,), dtype=torch.int32, device=device), "work": [], "g_out": None, "g_cnt": None}
            if self.world > 1 and self.rank == dst:
                slot["g_out"] = torch.zeros((self.world, self.mx, cap, 4), dtype=torch.int32, device=device)
                slot["g_cnt"] = torch.zeros((self.world, self.mx), dtype=torch.int32, device=device)
            self.slots.append(slot)

    def _wait(self, slot, host_sync=False):
        had = bool(slot["work"])
        for w in slot["work"]:
            w.wait()
        slot["work"] = []
        # RCCL's wait() only orders torch's current stream behind the collective; a producer that writes the slot
        # from its own streams (fmx_match_batch_keep) needs the host to know the gather is over
        if had and host_sync and slot["out"].is_cuda:
            self.torch.cuda.current_stream(slot["out"].device).synchronize()

    def buffers(self, i, host_sync=False):
        """(records, counts) tensors of step i for this rank's queries; safe to overwrite when returned."""
        slot = self.slots[i % len(self.slots)]
        self._wait(slot, host_sync)
        return slot["out"][:self.nq], slot["cnt"][:self.nq]

    def launch(self, i):
        """Starts the gather of step i's buffers (call after the step's kernels are enqueued)."""
        if self.world == 1:
            return
        slot = self.slots[i % len(self.slots)]
        on_dst = self.rank == self.dst
        if self.staged:  # through host memory, synchronously (dry runs only)
            for name, gname in (("out", "g_out"), ("cnt", "g_cnt")):
                src = slot[name].cpu()
                parts = [self.torch.empty_like(src) for _ in range(self.world)] if on_dst else None
                self.dist.gather(src, parts, dst=self.dst)
                if on_dst:
                    slot[gname].copy_(self.torch.stack(parts))
            return
        slot["work"] = [
            self.dist.gather(slot["out"], list(slot["g_out"].unbind(0)) if on_dst else None, dst=self.dst, async_op=True),
            self.dist.gather(slot["cnt"], list(slot["g_cnt"].unbind(0)) if on_dst else None, dst=self.dst, async_op=True)]

    def wait_all(self):
        for slot in self.slots:
            self._wait(slot)

    def result(self, i):
        """On `dst`: (records, counts) of step i in global query order (after its gather completed)."""
        slot = self.slots[i % len(self.slots)]
        self._wait(slot)
        if self.world == 1:
            return slot["out"][:self.nq], slot["cnt"][:self.nq]
        if self.rank != self.dst:
            return None, None
        t = self.torch
        return (t.cat([slot["g_out"][r, :self.sizes[r]] for r in range(self.world)]),
                t.cat([slot["g_cnt"][r, :self.sizes[r]] for r in range(self.world)]))
