#!/usr/bin/env python3
"""Side benchmarks for the BASELINE.json configs that are NOT the headline bench line
(bench.py measures configs[1]):  C3 (1M x 768 bf16, 256 queries, k=100, matrix-core path) and the
C5 scorer part (1M documents: I_hat row-cosine at d=512, robust fit, DEWI score).  Prints one JSON
object; run on the GPU box from the repo root:  python3 scripts/bench_configs.py
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"))
from dewi import _engine as eng  # noqa: E402
from dewi import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
lib = nat.load_library()
out = {}


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---------------------------------------------------------------- C3
n, d, b, k = 1_000_000, 768, 256, 100
g = torch.Generator(device=dev)
g.manual_seed(42)
emb = torch.randn((n, d), generator=g, device=dev)
emb /= emb.norm(dim=1, keepdim=True)
c32 = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine")
cb = c32.to_bf16()
Q = torch.randn((8, b, d), generator=g, device=dev)
it = iter(range(10**9))
t = timed(lambda: cb.search_device(Q[next(it) % 8], k, 0.3, 0.0), reps=40)
ids, _ = cb.search_device(Q[0], k, 0.3, 0.0)
out["C3_bf16_b256_k100"] = {"ms_per_batch": round(t * 1e3, 4), "queries_per_s": round(b / t, 1),
                            "hbm_GBps": round(n * d * 2 / t / 1e9, 1), "TFLOPs": round(2 * b * n * d / t / 1e12, 1),
                            "overflowed_queries": int((ids[:, 0] < 0).sum().item())}
t1 = timed(lambda: cb.search_device(Q[next(it) % 8][:1].contiguous(), 10, 0.3, 0.0), reps=200)
out["bf16_b1_k10"] = {"ms_per_query": round(t1 * 1e3, 4), "queries_per_s": round(1 / t1, 1),
                      "hbm_GBps": round(n * d * 2 / t1 / 1e9, 1)}
t4 = timed(lambda: c32.search_device(Q[next(it) % 8][:4].contiguous(), 10, 0.3, 0.0), reps=100)
out["f32_b4_k10"] = {"ms_per_batch": round(t4 * 1e3, 4), "queries_per_s": round(4 / t4, 1)}
Q8 = torch.randn((8, 8, d), generator=g, device=dev)
t8 = timed(lambda: c32.search_device(Q8[next(it) % 8], 10, 0.3, 0.0), reps=100)
out["f32_b8_k10"] = {"ms_per_batch": round(t8 * 1e3, 4), "queries_per_s": round(8 / t8, 1)}
del Q8
del cb, c32, emb, Q
torch.cuda.empty_cache()

# ---------------------------------------------------------------- C5 scorer part
n, d = 1_000_000, 512
te = torch.randn((n, d), generator=g, device=dev)
ie = te * 0.5 + torch.randn((n, d), generator=g, device=dev)
ih = torch.empty(n, device=dev)
t = timed(lambda: nat.check(lib.dewi_row_cosine_f32(nat.ptr(te), nat.ptr(ie), nat.ptr(ih), n, d, nat.stream_ptr())))
out["C5_I_hat_row_cosine"] = {"ms": round(t * 1e3, 4), "hbm_GBps": round(2 * n * d * 4 / t / 1e9, 1)}
from dewi.signals import redundancy_top1  # noqa: E402
nr = 262_144
t_h, i_h = te[:nr].cpu().numpy(), ie[:nr].cpu().numpy()
redundancy_top1(t_h[:4096], i_h[:4096])          # warm-up (library load, workspaces)
t0 = time.perf_counter()
red = redundancy_top1(t_h, i_h)                   # bf16 corpus + batched matrix-core path, host arrays in and out
t = time.perf_counter() - t0
out["C5_redundancy_top1_262144x512"] = {"s": round(t, 4), "self_join_TFLOPs": round(2 * nr * nr * d / t / 1e12, 1),
                                        "note": "host fp32 arrays in, host fp32 out (includes H2D of 2 x 0.5 GB, normalise, bf16 convert)"}
del t_h, i_h
if "--c5-full" in sys.argv:                       # the whole C5 corpus: 1 M text x 1 M image embeddings
    t_h, i_h = te.cpu().numpy(), ie.cpu().numpy()
    t0 = time.perf_counter()
    red = redundancy_top1(t_h, i_h)
    t = time.perf_counter() - t0
    out["C5_redundancy_top1_1Mx512"] = {"s": round(t, 3), "self_join_TFLOPs": round(2 * n * n * d / t / 1e12, 1),
                                        "note": "host fp32 arrays in, host fp32 out; 977 batches of 1024 queries"}
    del t_h, i_h
sig = torch.rand((7, n), generator=g, device=dev)
med = torch.empty(7, device=dev)
mad = torch.empty(7, device=dev)
wsb = int(lib.dewi_robust_fit_workspace_bytes(7))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
t = timed(lambda: nat.check(lib.dewi_robust_fit_f32(nat.ptr(sig), n, n, 7, nat.ptr(med), nat.ptr(mad), nat.ptr(ws), wsb,
                                                    nat.stream_ptr())))
out["C5_robust_fit_7x1M"] = {"ms": round(t * 1e3, 4), "algorithmic_GBps": round(2 * 3 * 7 * n * 4 / t / 1e9, 1)}
from dewi.sharded import HipFitSteps, ShardedRobustFit  # noqa: E402
fit1 = ShardedRobustFit(HipFitSteps(sig), n)      # world 1: the split entry points, 21 C-ABI calls + 2 D2H
t = timed(lambda: fit1.fit())
out["C5_robust_fit_7x1M_split_steps"] = {"ms": round(t * 1e3, 4)}
import ctypes
arr7 = ctypes.c_double * 7
m7 = arr7(*med.cpu().double().tolist())
d7 = arr7(*[x or 1e-8 for x in mad.cpu().double().tolist()])
w5 = (ctypes.c_double * 5)(1, 1, 1, 1, 1)
o64 = torch.empty(n, dtype=torch.float64, device=dev)
o32 = torch.empty(n, dtype=torch.float32, device=dev)
t = timed(lambda: nat.check(lib.dewi_score_f64(nat.ptr(sig), 0, n, n, m7, d7, w5, 3.0, 0, nat.ptr(o64), nat.ptr(o32),
                                               nat.stream_ptr())))
out["C5_score_1M"] = {"ms": round(t * 1e3, 4), "GBps": round((7 * 4 + 8 + 4) * n / t / 1e9, 1)}
print(json.dumps(out, indent=1))
