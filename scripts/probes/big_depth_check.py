"""Depth-split matrix-core pass at 8 M rows (24.6 GB fp32, then bf16): a 12-query batch against the row-per-wave
kernels one query at a time — 64-bit tile addressing, ~980 tiles per workgroup.  GPU box, repo root:
    python3 scripts/probes/big_depth_check.py
"""
import sys, torch
sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
n, d, k = 8_000_000, 768, 10
emb = torch.empty((n, d), device=dev)
for s in range(0, n, 1_000_000):
    emb[s:s + 1_000_000] = torch.randn((1_000_000, d), generator=g, device=dev)
emb /= emb.norm(dim=1, keepdim=True)
c = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine")
Q = torch.randn((12, d), generator=g, device=dev)
ids_b, sc_b = c.search_device(Q, k, 0.3, 0.0)            # depth-split pass over 8 M rows
same = 0
for j in range(12):
    i1, s1 = c.search_device(Q[j:j + 1], k, 0.3, 0.0)    # row-per-wave kernel
    same += int((i1[0] == ids_b[j]).sum())
    assert torch.allclose(s1[0].sort().values, sc_b[j].sort().values, atol=3e-6), j
print("8M rows: batch ids equal to single-query ids in", same, "of", 12 * k, "slots; max id", int(ids_b.max()))
cb = c.to_bf16(); del c, emb
ids_b, sc_b = cb.search_device(Q, k, 0.3, 0.0)
same = 0
for j in range(12):
    i1, s1 = cb.search_device(Q[j:j + 1], k, 0.3, 0.0)
    same += int((i1[0] == ids_b[j]).sum())
    assert torch.allclose(s1[0].sort().values, sc_b[j].sort().values, atol=3e-6), j
print("8M rows bf16: batch ids equal in", same, "of", 12 * k, "; max id", int(ids_b.max()))
