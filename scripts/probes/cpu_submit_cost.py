"""How much host time one step of the sharded throughput loop costs (forced RCCL path, world 1)."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 125000, 768, 10
g = torch.Generator(device=dev); g.manual_seed(1)
emb = torch.randn((n, d), generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
corpus = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), "cosine")
Q = torch.randn((256, 1, d), generator=g, device=dev)
c = 2 * k
fin = torch.cuda.Stream(); torch.cuda.set_stream(fin)
pipe = eng.PipelinedSearcher(corpus, k, 0.3, 0.0, n_queries=1, n_candidates=c, finish_stream=fin, depth=4,
                             scan_streams=int(os.environ.get("SS", "2")))
send = torch.empty((1, c, 4), dtype=torch.int32, device=dev); recv = torch.empty((1, 1, c, 4), dtype=torch.int32, device=dev)
oi = torch.empty((1, k), dtype=torch.int64, device=dev); osc = torch.empty((1, k), dtype=torch.float32, device=dev)
def step(j, parts):
    t = [time.perf_counter()]
    pipe.submit(Q[j % 256], out_records=send); t.append(time.perf_counter())
    w = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=True); t.append(time.perf_counter())
    w.wait(); t.append(time.perf_counter())
    eng.merge_rerank_device(recv, c, k, 0.3, 0.0, oi, osc); t.append(time.perf_counter())
    for i in range(4): parts[i] += t[i + 1] - t[i]
for rep in range(3):
    parts = [0.0] * 4
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 3000
    for j in range(N): step(j, parts)
    t_enq = time.perf_counter() - t0
    pipe.drain(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"rows {n}: enqueue {t_enq/N*1e6:.1f} us/step, total {t_all/N*1e6:.1f} us/step | submit {parts[0]/N*1e6:.1f} all_gather {parts[1]/N*1e6:.1f} wait {parts[2]/N*1e6:.1f} merge {parts[3]/N*1e6:.1f}")
dist.destroy_process_group()
