"""GPU: the multi-rank loop of bench.py — doc-id slicing, query groups sharing one all-gather, merge, conditioning
and warm-up with collectives inside, the sharded-result == single-GPU-result check and the latency leg — REHEARSED
with two ranks that share cuda:0 and exchange their records over gloo (DEWI_BENCH_BACKEND=gloo).  The driver's
8-GPU run is the first time this code meets RCCL with more than one rank; this test makes sure it is not also the
first time the code runs at world > 1.  Its throughput means nothing."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling,launcher", [("strong", "torchrun"), ("weak", "torchrun"), ("strong", "self")])
def test_bench_two_ranks_on_one_gpu(scaling, launcher):
    """launcher "torchrun": the driver's form (python -m torch.distributed.run ... bench.py --gpus 2).
    launcher "self": plain `python bench.py --gpus 2` — bench.py starts its two ranks itself, as a child process,
    before anything touches the GPU, and relays rank 0's JSON line and the exit code."""
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, DEWI_BENCH_BACKEND="gloo", DEWI_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(key, None)
    head = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", str(_free_port())] if launcher == "torchrun" else [sys.executable])
    cmd = head + [str(repo / "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "6",
           "--docs", "150001" if scaling == "strong" else "80000", "--condition-ms", "5", "--latency-queries", "12",
           "--scaling", scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(repo))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 24 and d["warmup"] == 6 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["parallelism"].startswith("REHEARSAL")
    assert d["rccl"]["backend"] == "gloo" and d["rccl"]["world"] == 2 and d["rccl"]["ranks_seen"] == [0, 1]
    assert sum(d["rccl"]["rows_per_rank"]) == d["config"]["docs"] and d["rccl"]["exchange"]["alone_p50_ms"] > 0
    if scaling == "strong":
        assert d["config"]["docs"] == 150001 and d["config"]["rows_per_gpu"] == 75000   # rank 0 of an uneven split
        assert d["sharded_parity"] == {"queries_checked": 16, "mismatches_vs_single_gpu": 0}
    else:
        assert d["config"]["docs"] == 160000 and d["config"]["rows_per_gpu"] == 80000
    assert d["p50_latency_ms"] > 0 and "cpu_baseline" not in d
