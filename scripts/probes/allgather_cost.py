import os, sys, time, torch
import torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
send = torch.zeros(80, dtype=torch.int32, device=dev); recv = torch.zeros(80, dtype=torch.int32, device=dev)
pg = dist.group.WORLD
def t(fn, n=3000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print("all_gather_into_tensor async+wait", round(t(lambda: dist.all_gather_into_tensor(recv, send, async_op=True).wait()), 1))
print("all_gather_into_tensor sync", round(t(lambda: dist.all_gather_into_tensor(recv, send)), 1))
try:
    print("pg._allgather_base", round(t(lambda: pg._allgather_base(recv, send)), 1))
    print("pg._allgather_base + wait", round(t(lambda: pg._allgather_base(recv, send).wait()), 1))
except Exception as e:
    print("pg._allgather_base failed", e)
try:
    import torch.distributed._functional_collectives as fc
    print("functional all_gather_tensor", round(t(lambda: fc.all_gather_tensor(send, 0, pg)), 1))
except Exception as e:
    print("functional failed", e)
ev = torch.cuda.Event(); s2 = torch.cuda.Stream()
print("event record + wait_event", round(t(lambda: (ev.record(), s2.wait_event(ev))), 1))
dist.destroy_process_group()
