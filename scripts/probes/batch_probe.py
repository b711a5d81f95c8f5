"""Batched queries on the matrix-core paths: whole-batch time and the event-timed filter pass at 1M x 768 for
several batch sizes.  Run on the GPU box from the repo root:
    [DEWI_HIP_LIB=<variant .so>] python3 scripts/probes/batch_probe.py [--bf16] [--l2] [--k K] [--graph] [--pipelined] [batch sizes...]
(fp32 corpus: csrc/knn_mfma_f32.hip from 5 queries; --bf16: depth pass up to 32 queries, 256-query kernel above.)
"""
import sys
import time

import torch

sys.path.insert(0, "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd")
from dewi import _engine as eng  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(42)
args = sys.argv[1:]
bf16 = "--bf16" in args
if bf16:
    args.remove("--bf16")
pipelined = "--pipelined" in args      # PipelinedSearcher, two scan streams: consecutive batches overlap their small kernels
if pipelined:
    args.remove("--pipelined")
space = "cosine"
if "--l2" in args:
    args.remove("--l2")
    space = "l2"
k = 10
if "--k" in args:
    i = args.index("--k")
    k = int(args[i + 1])
    del args[i:i + 2]
n, d = 1_000_000, 768
emb = torch.randn((n, d), generator=g, device=dev)
emb /= emb.norm(dim=1, keepdim=True)
c = eng.DeviceCorpus(emb, torch.rand(n, device=dev), torch.rand(n, device=dev), space)
if bf16:
    c = c.to_bf16()
    del emb
if space == "l2" and bf16:
    eng.tuning(0, 0, -1, 2)      # l2 over a bf16 corpus on the matrix cores is an opt-in (default: exact row kernels)
elem = 2 if bf16 else 4
args = [a for a in args if a not in ("--graph", "--split")]
sizes = [int(a) for a in args] or [8, 32, 64, 256]
for b in sizes:
    Q = torch.randn((8, b, d), generator=g, device=dev)
    for i in range(100):                 # ~50 ms of work first: shorter runs are measured while the clocks still settle
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    reps = 400
    t0 = time.perf_counter()
    for i in range(reps):
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    bursts = []
    for nb in (5, 20, 200):              # short bursts from an idle GPU run at a higher clock than a sustained stream
        torch.cuda.synchronize()
        time.sleep(0.05)
        t0 = time.perf_counter()
        for i in range(nb):
            c.search_device(Q[i % 8], k, 0.3, 0.0)
        torch.cuda.synchronize()
        bursts.append((nb, (time.perf_counter() - t0) / nb * 1e3))
    eng.timing(1)
    t0 = time.perf_counter()
    for i in range(20):
        c.search_device(Q[i % 8], k, 0.3, 0.0)
    torch.cuda.synchronize()
    t_timed = (time.perf_counter() - t0) / 20
    ms, cnt = eng.timing_read()
    eng.timing(0)
    ids, _ = c.search_device(Q[0], k, 0.3, 0.0)
    if "--split" in sys.argv:
        # the two halves of a batch on their own: dewi_knn_scan (prepare, sample pass, threshold, filter pass) and
        # dewi_knn_finish (select + re-rank), each as a back-to-back stream
        from dewi import _native as nat
        lib = nat.load_library()
        cc = min(2 * k, n)
        need = int(lib.dewi_knn_workspace_bytes(n, d, b, cc))
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        oi = torch.empty((b, k), dtype=torch.int64, device=dev)
        osc = torch.empty((b, k), dtype=torch.float32, device=dev)
        et, sp = (1 if bf16 else 0), nat.SPACE_CODES[space]
        def scan(i):
            nat.check(lib.dewi_knn_scan(nat.ptr(c.emb), et, n, d, nat.ptr(Q[i % 8]), b, cc, sp, nat.ptr(ws), need, nat.stream_ptr()))
        def fin():
            nat.check(lib.dewi_knn_finish(nat.ptr(ws), need, nat.ptr(c.emb), et, n, d, nat.ptr(Q[0]), b, cc, sp, k, 0.3, 0.0, nat.ptr(c.dewi32), nat.ptr(c.ent32), 0,
                                          nat.ptr(oi), nat.ptr(osc), 0, nat.stream_ptr()))
        for what, fn in (("scan", scan), ("finish", lambda i: fin()), ("scan+finish", lambda i: (scan(i), fin()))):
            for i in range(5):
                fn(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                fn(i)
            torch.cuda.synchronize()
            print(f"        {what}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms", flush=True)
    if "--graph" in sys.argv:
        # the batch's five kernels replayed from a captured graph: what the dependent-launch gaps cost
        oi = torch.empty((b, k), dtype=torch.int64, device=dev)
        osc = torch.empty((b, k), dtype=torch.float32, device=dev)
        qbuf = Q[0].clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            c.search_device(qbuf, k, 0.3, 0.0, oi, osc)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                c.search_device(qbuf, k, 0.3, 0.0, oi, osc)
            for i in range(5):
                graph.replay()
            side.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                graph.replay()
            side.synchronize()
            tg = (time.perf_counter() - t0) / reps
        same = bool(torch.equal(oi, ids))
        print(f"        graph replay: {tg * 1e3:.4f} ms per batch  ids equal {same}", flush=True)
    if pipelined:
        for streams in (1, 2, 3):
            ps = eng.PipelinedSearcher(c, k, 0.3, 0.0, n_queries=b, depth=4, scan_streams=streams)
            oi = torch.empty((8, b, k), dtype=torch.int64, device=dev)
            osc = torch.empty((8, b, k), dtype=torch.float32, device=dev)
            for i in range(8):
                ps.submit(Q[i % 8], oi[i % 8], osc[i % 8])
            ps.drain()
            t0 = time.perf_counter()
            for i in range(reps):
                ps.submit(Q[i % 8], oi[i % 8], osc[i % 8])
            ps.drain()
            tp = (time.perf_counter() - t0) / reps
            same = bool(torch.equal(oi[0], ids))
            print(f"        pipelined, {streams} scan stream(s): {tp * 1e3:.4f} ms per batch  ids equal {same}", flush=True)
    print(f"B={b:4d} batch {t * 1e3:.4f} ms  filter pass {ms:.4f} ms x {cnt // 20} per batch"
          f"  ({n * d * elem / ms / 1e6:.0f} GB/s)  refused {int((ids[:, 0] < 0).sum())}"
          f"  [batch with the events in: {t_timed * 1e3:.4f} ms]  bursts from idle: " + ", ".join(f"{nb}: {ms_:.4f}" for nb, ms_ in bursts),
          flush=True)
