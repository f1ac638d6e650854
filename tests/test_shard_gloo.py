"""World-size-2 gloo test (CPU) of the N > 1 plumbing: rank-sharded batches + the gather of top-N
records reproduce the single-process result.  The per-rank compute is the oracle here (no GPU in this
container); on the GPU box bench.py runs the same helpers over RCCL with the HIP path."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tools"))
import numpy as np, torch, torch.distributed as dist
import fmx_import, oracle_lib, workloads
shard = fmx_import.load().shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(3000, 301, 25)
o = workloads.build_oracle(c_off, c_tok, vs, nthreads=2)
prm = oracle_lib.Params(fuzzy=0.7, nmatch=5)
off, tok, lo, hi = shard.shard_batch(q_off, q_tok, rank, world)
out, cnt = o.match_batch(prm, off, tok, 5, nthreads=2)
t_out = torch.from_numpy(out.view(np.int32).reshape(len(cnt), 5, 4).copy())
t_cnt = torch.from_numpy(cnt.astype(np.int32))
g_out, g_cnt = shard.gather_records(t_out, t_cnt, dst=0)
# the persistent double-buffered gather bench.py uses: three "steps" through two slots, step i's
# gather still in flight while step i + 1 fills the other slot
rg = shard.RecordGather(len(cnt), 5, torch.device("cpu"), dst=0)
for i in range(3):
    b_out, b_cnt = rg.buffers(i)
    b_out.copy_(t_out + i); b_cnt.copy_(t_cnt)   # step i's results (records offset by i to tell steps apart)
    rg.launch(i)
rg.wait_all()
r_out, r_cnt = rg.result(2)
if rank == 0:
    assert torch.equal(r_out, g_out + 2) and torch.equal(r_cnt, g_cnt)
    r1, _ = rg.result(1)
    assert torch.equal(r1, g_out + 1)
else:
    assert r_out is None
# bench.py's shape: T + 1 slots for T steps in flight (here T = 4), eleven steps, a slot handed out again only when its gather is over;
# every step's assembled block checked on rank 0 before its slot is reused
T = 4
rg = shard.RecordGather(len(cnt), 5, torch.device("cpu"), dst=0, depth=T + 1)
assert rg.sizes == [shard.shard_range(301, r, world)[1] - shard.shard_range(301, r, world)[0] for r in range(world)]
for i in range(11):
    b_out, b_cnt = rg.buffers(i, host_sync=True)
    b_out.copy_(t_out + 100 + i); b_cnt.copy_(t_cnt)
    rg.launch(i)
    if i >= T:
        ro, rc = rg.result(i - T)
        if rank == 0:
            assert torch.equal(ro, g_out + 100 + i - T) and torch.equal(rc, g_cnt), i
        else:
            assert ro is None and rc is None
rg.wait_all()
for i in range(11 - T, 11):
    ro, rc = rg.result(i)
    if rank == 0:
        assert torch.equal(ro, g_out + 100 + i) and torch.equal(rc, g_cnt), i
if rank == 0:
    full_out, full_cnt = o.match_batch(prm, q_off, q_tok, 5, nthreads=2)
    assert g_out.shape[0] == len(full_cnt) == 301
    assert np.array_equal(g_cnt.numpy().astype(np.uint32), full_cnt)
    got = g_out.numpy().reshape(-1, 5 * 4)
    exp = full_out.view(np.int32).reshape(-1, 5 * 4)
    for q in range(301):
        n = int(full_cnt[q]) * 4
        assert np.array_equal(got[q, :n], exp[q, :n]), q
    print("SHARD_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_ranges(fmx):
    import fmx_import
    shard = fmx_import.load().shard
    for nq in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(nq, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nq
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


import pytest


@pytest.mark.parametrize("world,port", [(2, "29531"), (3, "29533")])
def test_world_size_n_gloo(tmp_path, world, port):
    """world 2: the contract's case; world 3: blocks of 100 / 100 / 101 queries (uneven three ways)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SHARD_OK" in r.stdout
