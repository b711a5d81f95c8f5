"""Times dewi_robust_fit_f32 at 7 x 1M and reports which columns took the whole-column fallback
(Counters.pad of csrc/robust_fit_fast.hip)."""
import sys
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"))
from dewi import _native as nat
lib = nat.load_library()
n, ns = 1_000_000, 7
rs = np.random.RandomState(1042)
sig = np.stack([rs.gamma(2, 0.5, n), rs.gamma(2, 0.5, n) * 1.5, rs.gamma(2, 0.3, n), rs.gamma(2, 0.3, n) * 1.5,
                rs.beta(2, 2, n), rs.beta(1, 5, n), rs.beta(1, 10, n)]).astype(np.float32)
S = torch.from_numpy(sig).cuda()
med = torch.empty(ns, dtype=torch.float32, device="cuda"); mad = torch.empty_like(med)
wsb = int(lib.dewi_robust_fit_workspace_bytes(ns)); ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
def fit():
    nat.check(lib.dewi_robust_fit_f32(nat.ptr(S), n, n, ns, nat.ptr(med), nat.ptr(mad), nat.ptr(ws), wsb, nat.stream_ptr()))
for _ in range(5): fit()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
for a, b in ev:
    a.record(); fit(); b.record()
torch.cuda.synchronize()
print("fit ms", np.mean([a.elapsed_time(b) for a, b in ev]))
fast_bytes = ((128 * 2 * ns + 255) // 256) * 256 + 4 * ns * 1024 * 1024 + 256
hist_bytes = wsb - fast_bytes
ctr = ws[hist_bytes: hist_bytes + 128 * 2 * ns].view(torch.int32).view(2, ns, 32).cpu().numpy()
print("counters [phase][col]: lt eqlo eqhi nan overflow ticket fallback pad | 16 bucket counts | stamps (10 ns): bracket stream publish tail")
np.set_printoptions(linewidth=250)
print(ctr[:, :, :8]); print(ctr[:, :, 8:24].sum(axis=2)); print(ctr[:, :, 24:28])
