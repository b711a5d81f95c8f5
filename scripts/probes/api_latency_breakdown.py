"""Where the blocking search() spends its time beyond the two kernels (1 M x 768, batch 1)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(42)
n, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 768
emb = torch.randn((n, d), generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
c = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine")
Q = np.random.RandomState(1).randn(300, d).astype(np.float32)
def med(f, n=200):
    xs = []
    for j in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(j); xs.append(time.perf_counter() - t)
    return np.median(xs[10:]) * 1e6
print("search() total            %.1f us" % med(lambda j: c.search(Q[j], 10, 0.3, 0.0)))
print("stage_queries + sync      %.1f us" % med(lambda j: (c.stage_queries(Q[j]), torch.cuda.synchronize())))
qd = c.stage_queries(Q[0]); oi = torch.empty((1, 10), dtype=torch.int64, device=dev); os_ = torch.empty((1, 10), device=dev)
print("search_device + sync      %.1f us" % med(lambda j: (c.search_device(qd, 10, 0.3, 0.0, oi, os_), torch.cuda.synchronize())))
print("search_device enqueue     %.1f us" % med(lambda j: c.search_device(qd, 10, 0.3, 0.0, oi, os_)))
hp = torch.empty((1, 10), dtype=torch.int64, pin_memory=True)
print("d2h 80 B pinned + sync    %.1f us" % med(lambda j: (hp.copy_(oi, non_blocking=True), torch.cuda.current_stream().synchronize())))
print("empty sync                %.1f us" % med(lambda j: torch.cuda.current_stream().synchronize()))
