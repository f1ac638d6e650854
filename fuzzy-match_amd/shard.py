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
