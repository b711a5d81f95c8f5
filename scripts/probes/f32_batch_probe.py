"""fp32 corpus, batched queries (matrix-core depth pass, csrc/knn_mfma_f32.hip): whole-batch time and the
event-timed filter pass at 1M x 768 for several batch sizes.  Run on the GPU box from the repo root:
    [DEWI_HIP_LIB=<variant .so>] python3 scripts/probes/f32_batch_probe.py [batch sizes...]
"""
import sys
import time

import torch

sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(42)
n, d, k = 1_000_000, 768, 10
emb = torch.randn((n, d), generator=g, device=dev)
emb /= emb.norm(dim=1, keepdim=True)
c = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine")
sizes = [int(a) for a in sys.argv[1:]] or [8, 32, 64, 256]
for b in sizes:
    Q = torch.randn((8, b, d), generator=g, device=dev)
    for i in range(5):
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    reps = 60
    t0 = time.perf_counter()
    for i in range(reps):
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    eng.timing(1)
    for i in range(20):
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    ms, cnt = eng.timing_read()
    eng.timing(0)
    ids, _ = c.search_device(Q[0], k, 0.3, 0.0)
    print(f"B={b:4d} batch {t * 1e3:.4f} ms  filter pass {ms:.4f} ms x {cnt // 20} per batch"
          f"  ({n * d * 4 / ms / 1e6:.0f} GB/s)  refused {int((ids[:, 0] < 0).sum())}", flush=True)
