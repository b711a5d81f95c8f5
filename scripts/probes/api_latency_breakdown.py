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
# zero-copy results: the select kernel writes ids / scores straight into pinned host memory (device-visible under unified addressing)
import ctypes
from dewi import _native as nat
hb = torch.empty(1 * 10 * 12, dtype=torch.uint8, pin_memory=True)
h_ids = hb[:80].view(torch.int64).view(1, 10); h_sc = hb[80:].view(torch.float32).view(1, 10)
lib = nat.load_library()
ws = c._workspace(1, 20)
def zc(j):
    rc = lib.dewi_knn_rerank_f32(nat.ptr(c.emb), c.n_rows, c.dim, nat.ptr(qd), 1, nat.ptr(c.dewi32), nat.ptr(c.ent32), 10, 0.3, 0.0, 0,
                                 h_ids.data_ptr(), h_sc.data_ptr(), nat.ptr(ws), ws.numel(), nat.stream_ptr())
    assert rc == 0
    torch.cuda.current_stream().synchronize()
print("search (pinned outputs) + sync %.1f us" % med(zc))
c.search_device(qd, 10, 0.3, 0.0, oi, os_); torch.cuda.synchronize()
print("zero-copy result equal:", bool((h_ids == oi.cpu()).all()) and bool((h_sc == os_.cpu()).all()))
def plain(j):
    c.search_device(qd, 10, 0.3, 0.0, oi, os_); hp.copy_(oi, non_blocking=True); torch.cuda.current_stream().synchronize()
print("search_device + d2h + sync     %.1f us" % med(plain))
