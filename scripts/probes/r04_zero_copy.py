"""Blocking one-query search latency: results through a device buffer + D2H copy (as shipped) against results written by
the select kernel straight into pinned host memory (no copy command).  GPU box, repo root."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng  # noqa: E402

for n in (10_000, 1_000_000):
    d, k = 768, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    emb = torch.randn((n, d), generator=g, device="cuda")
    emb /= emb.norm(dim=1, keepdim=True)
    c = eng.DeviceCorpus(emb, torch.rand(n, device="cuda"), torch.rand(n, device="cuda"), "cosine")
    Q = np.random.RandomState(0).randn(256, d).astype(np.float32)
    for mode in ("copy", "pinned", "copy", "pinned"):
        ids_p = torch.empty((1, k), dtype=torch.int64, pin_memory=True)
        sc_p = torch.empty((1, k), dtype=torch.float32, pin_memory=True)
        lat = []
        for j in range(300):
            t0 = time.perf_counter()
            if mode == "copy":
                ids, sc = c.search(Q[j % 256], k, 0.3, 0.0)
            else:
                q = c.stage_queries(Q[j % 256])
                c.search_device(q, k, 0.3, 0.0, ids_p, sc_p)
                torch.cuda.current_stream().synchronize()
                ids, sc = ids_p.numpy().copy(), sc_p.numpy().copy()
            lat.append(time.perf_counter() - t0)
        ref_ids, ref_sc = c.search(Q[(299) % 256], k, 0.3, 0.0)
        ok = np.array_equal(ids, ref_ids) and np.array_equal(sc, ref_sc)
        print(f"n={n} {mode}: p50 {np.median(lat[50:]) * 1e3:.4f} ms  min {min(lat) * 1e3:.4f}  equal={ok}", flush=True)
