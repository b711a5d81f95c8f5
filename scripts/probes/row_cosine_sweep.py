"""Timing and torch check of the I_hat row-cosine kernel (1 M x 512 x 2 matrices)."""
import os, sys, time, torch
sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _native as nat
lib = nat.load_library()
dev = torch.device("cuda:0")
n, d = 1_000_000, 512
g = torch.Generator(device=dev); g.manual_seed(1)
a = torch.randn((n, d), generator=g, device=dev); b = torch.randn((n, d), generator=g, device=dev)
o = torch.empty(n, device=dev)
def run():
    nat.check(lib.dewi_row_cosine_f32(nat.ptr(a), nat.ptr(b), nat.ptr(o), n, d, nat.stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): run()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
ref = torch.nn.functional.cosine_similarity(a[:4096], b[:4096])
print(f"{t*1e3:.4f} ms {2*n*d*4/t/1e9:.0f} GB/s maxerr {float((o[:4096]-ref).abs().max()):.2e}")
