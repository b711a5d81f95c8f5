#!/usr/bin/env python3
"""Interleaved A/B of scan launch shapes in ONE process (cdna_hip_programming.md §5.4 rule 24).

usage (GPU box, repo root): python3 scripts/tune_scan.py [--docs N] [--dim D] [--k K] [--rounds R] cfg ...
  cfg = blocks:rows_per_iter:nt   e.g. 256:8:1 512:4:1 0:0:-1
Prints, per configuration, the median / min scan-kernel time (hipEvents inside the library) and the
median whole-step time over the rounds.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"))
from dewi import _engine as eng  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--no-timing", action="store_true", help="do not bracket scans with hipEvents (step time only)")
ap.add_argument("cfgs", nargs="+")
a = ap.parse_args()

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(42)
emb = torch.randn((a.docs, a.dim), generator=g, device=dev)
emb /= emb.norm(dim=1, keepdim=True)
corpus = eng.DeviceCorpus(emb, torch.rand(a.docs, device=dev), torch.rand(a.docs, device=dev), "cosine")
if a.bf16:
    corpus = corpus.to_bf16()
Q = torch.randn((64, a.batch, a.dim), generator=g, device=dev)
cfgs = [tuple(int(x) for x in c.split(":")) for c in a.cfgs]
res = {c: {"scan": [], "step": []} for c in cfgs}
for rnd in range(a.rounds + 1):
    for c in cfgs:
        eng.tuning(*c)
        for j in range(10):
            corpus.search_device(Q[j], a.k, 0.3, 0.0)
        torch.cuda.synchronize()
        eng.timing(not a.no_timing)
        t0 = time.perf_counter()
        for j in range(a.steps):
            corpus.search_device(Q[j % 64], a.k, 0.3, 0.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        ms, n = eng.timing_read() if not a.no_timing else (0.0, 0)
        eng.timing(False)
        if rnd:  # round 0 is warm-up
            res[c]["scan"].append(ms)
            res[c]["step"].append(dt)
bytes_ = corpus.corpus_bytes()
print(f"corpus {a.docs}x{a.dim} {'bf16' if a.bf16 else 'fp32'} = {bytes_/1e9:.3f} GB, batch {a.batch}, k {a.k}")
for c in cfgs:
    s, st = np.array(res[c]["scan"]), np.array(res[c]["step"])
    print(f"blocks={c[0]:5d} R={c[1]} nt={c[2]:2d}  scan median {np.median(s):.4f} ms min {s.min():.4f}  "
          f"({bytes_/np.median(s)/1e6:.0f} GB/s)  step median {np.median(st):.4f} ms  overhead {np.median(st)-np.median(s):.4f}")
