"""bench.py's N > 1 control flow (rank sharding, text batches with keep + gather slots, strong and weak measurements, per-rank rates, the
one JSON line on rank 0) run end to end on a box with ONE GPU: two ranks on device 0, the collectives over gloo through
host memory (FMX_BENCH_BACKEND=gloo).  The driver's multi-GPU runs use RCCL with one GPU per rank; this only makes sure
that nothing on that path is broken."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu(gpu_device):
    env = dict(os.environ, FMX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("FMX_SMALL_PATH", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--sentences", "200000", "--queries", "40000", "--no-cpu-baseline", "--model-queries", "2000"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    # N > 1 defaults to BASELINE's shape: ONE batch of --queries cut over the ranks (strong); the weak shape is measured beside it
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["queries_per_step_all_ranks"] == 40000
    assert d["scaling_other"]["scaling"] == "weak" and d["scaling_other"]["queries_per_step_all_ranks"] == 80000
    assert len(d["per_rank"]) == 2 and d["parity"]["mismatches"] == 0 and d["parity"]["checked"] > 0 and d["value"] > 0
    assert "fmx_match_batch_text" in d["config"]["timed_path"] and d["device_resident"]["value"] > 0
    assert d["scaling_other"]["value"] > 0 and d["scaling_other"]["device_resident_value"] > 0
    # what the gather assembled on rank 0 from EVERY rank's block is checked against the oracle (input order), and the HBM traffic of
    # the dominant kernel is measured at N > 1 too (rank 0's rocprofv3 child passes)
    prb = d["parity"]["per_rank_block"]
    assert len(prb) == 2 and [x["rank"] for x in prb] == [0, 1]
    assert all(x["mismatches"] == 0 and x["checked"] > 0 and x["queries_in_block"] == 20000 for x in prb)
    assert isinstance(d["roofline"]["traffic"], int) and d["roofline"]["traffic"] > 0
