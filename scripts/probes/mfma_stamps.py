import sys, torch
sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(42)
n, d, b, k = 1_000_000, 768, 256, 100
emb = torch.randn((n, d), generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
cb = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine").to_bf16()
Q = torch.randn((b, d), generator=g, device=dev)
for _ in range(2):
    cb.search_device(Q, k, 0.3, 0.0)
    torch.cuda.synchronize()
