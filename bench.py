#!/usr/bin/env python3
"""Headline benchmark of the DEWI hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

Default (``--config c2``, the configuration BASELINE.json's metric is quoted on): queries/s (plus
p50 latency) of brute-force cosine kNN + DEWI re-rank over a 1M x 768 fp32 corpus, query batch 1,
k=10, eta=0.3 (configs[1]).  One "step" = one query through the whole hot path: corpus scan with
fused top-2k, select, blend, top-k.  Corpus, payload columns and queries are resident in HBM before
the timed region; results stay on the device (the PCIe-inclusive API latency is reported separately
as p50_latency_ms).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the 1M-row corpus is sharded by
contiguous doc-id range (strong scaling; ``--scaling weak`` keeps 1M rows per GPU = configs[3]);
each step scans the local shard, all-gathers the per-shard top-2k records (16 B each) and merges
them on every rank.  The all-gather of query i overlaps the scan of query i+1.

Other configurations of BASELINE.json, one JSON line each, same harness (1 GPU):
  --config c3   1M x 768 bf16, 256 queries per step, k=100: the batched matrix-core path
                (roofline carries both the HBM and the MFMA fraction of the filter-pass kernel);
  --config c4   configs[3] replayed on ONE GPU: 8 resident 1M-row fp32 shards, per step every shard
                is scanned (dewi_knn_candidates, global ids) and the records are merged
                (dewi_merge_rerank) — the whole exchange minus the wire;
  --config c5   1M documents, d=512: robust fit + DEWI score of the 7 signals and the I_hat
                row-cosine of the text/image embedding pair (the on-GPU part of configs[4]).

The JSON line also carries
  roofline      achieved HBM GB/s of the dominant kernel = algorithmic bytes / its mean duration,
                measured with hipEvents on the launch stream over >= 200 launches (a dedicated leg
                right after the timed region, plus the sampled launches of the timed region itself)
                against the 8 TB/s peak; `traffic` = HBM bytes per launch from the PMC counters of
                profiles/hbm_traffic.json, reported only when that file was recorded on these very
                sources (hash of csrc/ + include/);
  cpu_baseline  the NumPy oracle (a port of the reference's ExactIndex.search) timed on the host
                cores of this box on a bounded sample of the same workload (>= 50 queries whatever
                --steps is), used at the same time as the parity gate for the GPU results (rank 0,
                N=1 only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
PKG_DIR = REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"
sys.path.insert(0, str(PKG_DIR))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
MIN_ROOFLINE_LAUNCHES = 200
METRIC = "queries/sec + p50 latency, 1M×768 corpus, k=10, η=0.3, at 1/2/4/8 GPUs"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--docs", type=int, default=1_000_000, help="corpus rows (total for strong, per GPU for weak)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--eta", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=None, help="queries per step")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--shadow", type=int, default=-1,
                    help="bf16 shadow copy of the fp32 corpus for query batches (pre-selection + exact re-scoring): 1 on, 0 off, "
                         "-1 = on for --batch > 32")
    ap.add_argument("--emulate-shards", type=int, default=8, help="--config c4: resident shards on the one GPU")
    ap.add_argument("--cpu-queries", type=int, default=512,
                    help="queries timed on the CPU oracle: ~10 s of CPU work at C2, cut off after 30 s (0 = skip; at least 50 are run)")
    ap.add_argument("--latency-queries", type=int, default=200)
    ap.add_argument("--condition-ms", type=float, default=200.0,
                    help="untimed scanning before the warm-up steps, so that the timed region starts on settled clocks")
    ap.add_argument("--scan-blocks", type=int, default=0)
    ap.add_argument("--rows-per-iter", type=int, default=0)
    ap.add_argument("--nontemporal", type=int, default=-1)
    a = ap.parse_args()
    defaults = {"c2": dict(steps=1000, warmup=50, dim=768, k=10, batch=1),
                # c3: 100 + 500 batches = 0.24 s.  A 100-batch run (40 ms) ends before the chip has settled: 0.411 ms per
                # batch against 0.394 at 400 and at 1000 steps on the same box (same mean kernel time; DVFS, not the code)
                "c3": dict(steps=500, warmup=100, dim=768, k=100, batch=256),
                "c4": dict(steps=100, warmup=10, dim=768, k=10, batch=1),
                "c5": dict(steps=50, warmup=5, dim=512, k=10, batch=1)}[a.config]
    for key, val in defaults.items():
        if getattr(a, key) is None:
            setattr(a, key, val)
    return a


def make_corpus(torch, n_rows, dim, seed, device):
    """Unit-norm gaussian rows + payload columns, generated on the GPU (synthetic data with the
    distributions of the reference's harness, scripts/profile_index.py:49-70)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    emb = torch.empty((n_rows, dim), dtype=torch.float32, device=device)
    chunk = 131072
    for s in range(0, n_rows, chunk):
        e = min(n_rows, s + chunk)
        emb[s:e] = torch.randn((e - s, dim), generator=g, device=device, dtype=torch.float32)
    rs = np.random.RandomState(seed + 1000)
    cols = {
        "dewi": np.clip(rs.beta(2, 2, n_rows), 0, 1),
        "ht_mean": rs.gamma(2, 0.5, n_rows),
        "hi_mean": rs.gamma(2, 0.3, n_rows),
    }
    return emb, cols


def sources_digest() -> str:
    """sha256 over the kernel sources and the C header: what a PMC record must have been taken on."""
    h = hashlib.sha256()
    files = sorted((PKG_DIR / "csrc").glob("*")) + sorted((REPO / "include").glob("*.h"))
    for f in files:
        if f.is_file() and f.suffix in (".hip", ".hpp", ".cpp", ".h") or f.name == "Makefile":
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()


def recorded_traffic(key: str, kernel: str):
    """(bytes per launch or None, note).  profiles/hbm_traffic.json = {"sources_sha256", "commit",
    "records": {key: {"kernel", "bytes_per_launch", ...}}} written by scripts/summarize_pmc.py."""
    tfile = REPO / "profiles" / "hbm_traffic.json"
    if not tfile.exists():
        return None, "no PMC record (profiles/hbm_traffic.json missing)"
    try:
        rec = json.loads(tfile.read_text())
        if rec.get("sources_sha256") != sources_digest():
            return None, f"PMC record of commit {rec.get('commit', '?')} was taken on different kernel sources: not reported"
        ent = rec.get("records", {}).get(key)
        if not ent or ent.get("kernel", "").replace(" ", "") != kernel.replace(" ", ""):
            return None, f"no PMC record for {key} / {kernel}"
        return ent["bytes_per_launch"], f"rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction), commit {rec.get('commit', '?')}"
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable PMC record: {e}"


def device_payload(torch, nat, cols, n, device):
    c64 = [torch.from_numpy(np.ascontiguousarray(cols[k], dtype=np.float64)).to(device) for k in ("dewi", "ht_mean", "hi_mean")]
    dewi32 = torch.empty(n, dtype=torch.float32, device=device)
    ent32 = torch.empty(n, dtype=torch.float32, device=device)
    nat.check(nat.load_library().dewi_payload_soa_f64(nat.ptr(c64[0]), nat.ptr(c64[1]), nat.ptr(c64[2]), nat.ptr(dewi32),
                                                      nat.ptr(ent32), n, nat.stream_ptr()))
    return dewi32, ent32


def host_facts():
    """What the host gives this job: logical CPUs visible / in the affinity mask, the cgroup CPU quota if any, CPU model."""
    facts = {"os_cpu_count": os.cpu_count()}
    try:
        facts["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                facts["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:  # noqa: BLE001
        pass
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            facts["cgroup_cpu_quota"] = open(f).read().strip()
            break
        except Exception:  # noqa: BLE001
            continue
    # cgroup v2 "cpu.max" = "<quota us> <period us>" (or "max ..."): the CPUs' worth of time this job may use, however
    # many logical CPUs it can be scheduled on — threads beyond it only get throttled
    try:
        quota, period = facts.get("cgroup_cpu_quota", "").split()[:2]
        if quota != "max":
            facts["cpus_worth_of_quota"] = round(int(quota) / int(period), 2)
    except Exception:  # noqa: BLE001
        pass
    return facts


def host_copy_first_touched_in_parallel(t_dev, workers=None, rows_per_chunk=4096):
    """Device matrix -> host ndarray whose pages were FIRST TOUCHED by many threads (chunks of rows handed round-robin to
    a thread pool), so that on a multi-socket / multi-NUMA host the matrix is spread over every node's memory instead of
    sitting on the node of the one thread that copied it (which starves BLAS threads on the other nodes: the round-2
    probe found 8 threads fastest on a 256-CPU box for that reason)."""
    from concurrent.futures import ThreadPoolExecutor
    src = t_dev.cpu().numpy()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = workers or max(1, min(64, avail))
    dst = np.empty_like(src)                     # untouched pages
    n = src.shape[0]
    chunks = [(lo, min(n, lo + rows_per_chunk)) for lo in range(0, n, rows_per_chunk)]

    def work(w):
        for lo, hi in chunks[w::workers]:
            dst[lo:hi] = src[lo:hi]              # numpy releases the GIL for the copy
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(work, range(workers)))
    return dst


def best_blas_threads(fn):
    """The box may expose many more logical CPUs than this job's share; OpenBLAS then oversubscribes and
    slows down.  Probe a ladder of BLAS thread counts with fn() and return (fastest count, probe table, visible)."""
    from threadpoolctl import threadpool_limits
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    cands = sorted({t for t in (8, 16, 24, 32, 48, 64, 96, 128, 192, avail) if t <= avail})
    probe = {}
    for t in cands:
        with threadpool_limits(limits=t, user_api="blas"):
            fn()
            t1 = time.perf_counter()
            for _ in range(3):
                fn()
            probe[t] = (time.perf_counter() - t1) / 3
    return min(probe, key=probe.get), probe, avail


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract of this script
    is ONE JSON line on stdout, so file descriptor 1 points at stderr while the process group is brought up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def oracle_imports():
    sys.path.insert(0, str(REPO / "oracle"))
    sys.path.insert(0, str(REPO / "tests"))
    import dewi_oracle as orc           # checker / reported baseline only
    from parity import compare_query
    return orc, compare_query


# ======================================================================================================
# c2 (default) and its multi-GPU forms
# ======================================================================================================
def run_c2(args, torch, dist, eng, nat, rank, world, device):
    full = full_cols = None
    if args.scaling == "strong":
        total_rows = args.docs
        from dewi.sharded import shard_bounds
        lo, hi = shard_bounds(total_rows, world)[rank]        # balanced, even boundaries (see its docstring)
        # every rank generates the same full stream and keeps its slice: identical to the 1-GPU corpus
        full, full_cols = make_corpus(torch, total_rows, args.dim, 42, device)
        emb_raw = full[lo:hi].clone() if world > 1 else full
        cols = {k: v[lo:hi] for k, v in full_cols.items()}
        if world == 1 or rank != 0 or args.scaling != "strong":
            full = None      # rank 0 of a sharded run keeps the whole corpus for the post-run check
    else:
        total_rows = args.docs * world
        lo, hi = args.docs * rank, args.docs * (rank + 1)
        emb_raw, cols = make_corpus(torch, args.docs, args.dim, 42 + rank, device)
    n_local = hi - lo
    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(emb_raw), nat.ptr(emb_raw), n_local, args.dim,
                                                         nat.stream_ptr()))
    dewi32, ent32 = device_payload(torch, nat, cols, n_local, device)
    corpus = eng.DeviceCorpus(emb_raw, dewi32, ent32, "cosine", id_offset=lo)
    n_q = args.warmup + args.steps
    B = args.batch
    # (the headline line — no flags, one query per step — never takes the shadow: it stays the plain fp32 scan over N*d*4 bytes)
    shadowed = (B > 32 if args.shadow < 0 else args.shadow == 1) and world == 1 and args.dim % 32 == 0 and 160 <= args.dim <= 1536 and \
        min(2 * args.k, total_rows) <= (512 if B > 32 else 256)
    if shadowed:
        # query batches over the fp32 corpus: a matrix-core pass over a bf16 shadow copy as a pre-selection, candidates
        # re-scored from the fp32 rows (dewi_knn_rerank_f32_shadow): fp32-exact results, +50 % memory.  --shadow 1 with
        # one query per step: the one-query search through the shadow (an extra line, never the default)
        corpus.enable_bf16_shadow(single_query=(B == 1))
    qg = torch.Generator(device=device)
    qg.manual_seed(7)
    n_cpu = max(50, args.cpu_queries) if args.cpu_queries > 0 else 0
    n_distinct = min(max(n_q, MIN_ROOFLINE_LAUNCHES + 8, n_cpu), 4096)
    Q = torch.randn((n_distinct, B, args.dim), generator=qg, device=device, dtype=torch.float32)
    k, eta = args.k, args.eta
    c = min(2 * k, total_rows)
    out_ids = torch.empty((n_distinct, B, k), dtype=torch.int64, device=device)
    out_sc = torch.empty((n_distinct, B, k), dtype=torch.float32, device=device)
    torch.cuda.synchronize()

    # ------------------------------------------------------------------ the step
    # DEWI_BENCH_FORCE_DIST=1 runs the sharded code path (RCCL all-gather + merge) even at world 1,
    # so that it can be exercised on a single-GPU box.
    force_dist = os.environ.get("DEWI_BENCH_FORCE_DIST", "0") == "1"
    if force_dist and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with stdout_to_stderr():
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
            dist.barrier()                                   # brings the communicator (and RCCL's banner) up here
    sharded = world > 1 or force_dist
    # Single GPU: queries back to back on ONE stream (scan, select; next query).  Overlapping the select
    # of query i with the scan of query i+1 on a second stream (DEWI_BENCH_PIPELINE=1) was measured
    # SLOWER on MI355X (0.4506 vs 0.4434 ms/step at 1M rows, 84.7 vs 77.3 us at 125K): the cross-stream
    # event waits cost more than the 9 us select they hide.
    serial = os.environ.get("DEWI_BENCH_PIPELINE", "0") != "1"
    qs = [Q[j] for j in range(n_distinct)]
    oi = [out_ids[j] for j in range(n_distinct)]
    osc = [out_sc[j] for j in range(n_distinct)]
    G = 1
    if not sharded and serial:
        def run(first, count):
            for i in range(first, first + count):
                j = i % n_distinct
                corpus.search_device(qs[j], k, eta, 0.0, oi[j], osc[j])
    elif not sharded:
        # Throughput loop: scans back to back on one stream; the select/blend/top-k of query i runs on
        # a second stream and overlaps the scan of query i+1 (two workspaces in rotation).
        pipe = eng.PipelinedSearcher(corpus, k, eta, 0.0, n_queries=B)

        def run(first, count):
            for i in range(first, first + count):
                j = i % n_distinct
                pipe.submit(qs[j], oi[j], osc[j])
            pipe.drain()
    else:
        # Sharded throughput loop.  Scans alternate between two scan streams (the tail of one overlaps the
        # ramp of the next: at 125 K rows per GPU a lone scan spends a quarter of its 68 us ramping up and
        # merging).  Candidate records of G consecutive queries share ONE all-gather and ONE merge launch:
        # a torch.distributed collective costs ~26 us of host time however small it is, and with 125 K rows
        # per GPU the host, not the GPU, was pacing the loop (65 us of submission per query).  Up to
        # depth-1 groups are in flight behind the scans.
        depth = 3
        nbuf = depth + 1
        G = max(1, int(os.environ.get("DEWI_BENCH_GROUP", "4")))
        n_distinct -= n_distinct % G          # groups never wrap around the query ring
        fin = torch.cuda.Stream()
        torch.cuda.set_stream(fin)          # torch.distributed orders collectives against the current stream
        pipe = eng.PipelinedSearcher(corpus, k, eta, 0.0, n_queries=B, n_candidates=c, finish_stream=fin,
                                     depth=max(int(os.environ.get("DEWI_BENCH_WS_DEPTH", "4")), 2 * G),
                                     scan_streams=int(os.environ.get("DEWI_BENCH_SCAN_STREAMS", "2")))
        send = [torch.empty((G, B, c, 4), dtype=torch.int32, device=device) for _ in range(nbuf)]
        recv = [torch.empty((world * G * B * c * 4,), dtype=torch.int32, device=device) for _ in range(nbuf)]
        from collections import deque
        rehearsal = dist.get_backend() != "nccl"     # DEWI_BENCH_BACKEND=gloo: the loop's logic on ranks that share a GPU

        class _Done:
            def wait(self):
                return True

        def gather_records(dst, src):
            if not rehearsal:
                return dist.all_gather_into_tensor(dst, src, async_op=True)
            h_src = src.cpu()                          # synchronises the finish stream; staged through the host
            h_dst = torch.empty(dst.numel(), dtype=dst.dtype)
            dist.all_gather_into_tensor(h_dst, h_src)
            dst.copy_(h_dst)
            return _Done()

        def finish_group(work, j0, g, s):
            work.wait()
            lists = recv[s][: world * g * B * c * 4].view(world, g * B, c, 4)
            eng.merge_rerank_device(lists, c, k, eta, 0.0, out_ids[j0:j0 + g].view(g * B, k), out_sc[j0:j0 + g].view(g * B, k))

        def run(first, count):
            inflight = deque()
            i, n_group = first, 0
            while i < first + count:
                j0 = i % n_distinct
                g = min(G, first + count - i, n_distinct - j0)
                s = n_group % nbuf
                for u in range(g):
                    pipe.submit(qs[j0 + u], out_records=send[s][u])
                work = gather_records(recv[s][: world * g * B * c * 4], send[s][:g].view(-1))
                inflight.append((work, j0, g, s))
                if len(inflight) >= depth:
                    finish_group(*inflight.popleft())
                i += g
                n_group += 1
            while inflight:
                finish_group(*inflight.popleft())
            pipe.drain()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # the library's event pool is filled during the warm-up steps (every scan bracketed, readings discarded), so that
    # no hipEventCreate — ~0.1 ms of host time each — falls into the timed region (a 20-step run lost 0.5 ms to them)
    # Device conditioning, outside both the W warm-up steps and the timed region: after the (mostly host-paced) set-up
    # the chip needs ~150 ms of sustained scanning before its clocks sit where a long run has them — the first scans
    # take 0.447 ms, the 400th 0.430 (1 M x 768; measured with --steps 10..160 and with this loop off / 100 / 400
    # scans long).  A 20-step run measured from a cold chip reads 3 % low for that reason alone.  The same number of
    # steps on every rank (the sharded loop has collectives in it).
    est_step_s = max(args.docs * args.dim * 4 / max(world if args.scaling == "strong" else 1, 1) / 7.0e12, 30e-6)
    n_condition = int(args.condition_ms * 1e-3 / est_step_s)
    n_condition -= n_condition % G
    if n_condition > 0:
        run(0, n_condition)
    eng.timing(1)
    run(0, args.warmup)
    barrier()
    eng.timing_read()
    # hipEvents around every 8th scan kernel inside the timed region (the two event records cost ~5-10 us of
    # stream time, so the timed region is only sampled; the roofline leg below brackets every launch)
    eng.timing(8)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms, region_launches = eng.timing_read()
    eng.timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Roofline leg: the same loop body, every scan bracketed, at least MIN_ROOFLINE_LAUNCHES launches whatever
    # --steps was (>= 200 event-timed launches).  Outside the timed region: does not touch `value`.
    n_leg = max(MIN_ROOFLINE_LAUNCHES, 0)
    n_leg += (-n_leg) % G
    eng.timing(1)
    run(0, n_leg)
    barrier()
    leg_ms, leg_launches = eng.timing_read()
    eng.timing(False)
    tot_launches = region_launches + leg_launches
    scan_ms = (region_ms * region_launches + leg_ms * leg_launches) / tot_launches if tot_launches else 0.0

    qps = args.steps * B / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    elem = 2 if shadowed else corpus.emb.element_size()              # the dominant pass streams the bf16 shadow
    algo_bytes = n_local * args.dim * elem + B * args.dim * (2 if shadowed else 4)           # per scan launch, this rank
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # --batch >= 5 over the fp32 corpus: the depth-split matrix-core pass is the dominant kernel (one per 32 queries);
    # --batch > 32: the 256-query pass over the bf16 shadow (one per 256 queries)
    # --batch 1 --shadow 1 (cut of at most 32 rows): the bf16 ROW kernel over the shadow
    # (the library names the row kernel of this width itself: the tuned dim = 256 U kernels, the any-width ones, the generic)
    depth_args = f"{-(-args.dim // 256)},false,false,{'true' if args.dim % 256 else 'false'}"   # chunks, SAMPLE, L2, partial last chunk
    lib_name = corpus.scan_kernel_name(B, k)       # what the library's plan says streams the corpus for this shape
    kernel = ((f"mfma_scan_f32<false,{depth_args}>" if lib_name.startswith("mfma_scan_f32") else lib_name) if not shadowed
              else f"mfma_scan_bf16_s16<{args.dim // 16},false>" if (B > 32 and args.dim % 128 == 0 and args.dim <= 768)
              else f"scan_rows_bf16<{args.dim // 256},1,0,1,true>" if (B == 1 and c <= 32 and args.dim <= 1024 and args.dim % 256 == 0)   # (no tuned bf16 row kernel at 1536 / other widths)
              else f"mfma_scan_f32<true,{depth_args}>")
    traffic, traffic_note = recorded_traffic(f"{n_local}x{args.dim}x{elem}xB{B}", kernel)
    if traffic is None and shadowed and B < 32 and kernel.startswith("mfma_scan_f32"):
        # the depth-split pass is one launch of the same grid over the same rows for 1..32 active queries: the counter record
        # taken with 32 stands for the smaller batches (and says so)
        traffic, traffic_note = recorded_traffic(f"{n_local}x{args.dim}x{elem}xB32", kernel)
        if traffic is not None:
            traffic_note += f"; recorded with 32 queries per pass (same launch, {B} active here)"

    result = {
        "metric": METRIC,
        "value": round(qps, 2),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{total_rows} docs x d={args.dim} fp32, query batch={B}, k={k}, eta={eta}, "
                               f"brute-force cosine kNN + DEWI re-rank (BASELINE.json configs[1])",
                   "docs": total_rows, "dim": args.dim, "k": k, "eta": eta, "batch": B, "candidates": c,
                   "parallelism": (f"doc-id shards x{world} + RCCL all-gather" if not sharded or dist.get_backend() == "nccl" else
                                   f"REHEARSAL: {world} ranks on shared GPUs, records staged through the host ({dist.get_backend()})")
                   if sharded else "single GPU",
                   "bf16_shadow": shadowed,
                   "queries_in_flight": 1 if (serial and not sharded) else (3 * G if sharded else 2),
                   "rows_per_gpu": n_local, "conditioning_steps_before_warmup": n_condition},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": algo_bytes, "mean_kernel_ms": round(scan_ms, 5),
                     "launches_timed": tot_launches,
                     "timed_region": {"mean_kernel_ms": round(region_ms, 5), "launches": region_launches},
                     "roofline_leg": {"mean_kernel_ms": round(leg_ms, 5), "launches": leg_launches}},
    }

    if sharded:
        # Two scans are in flight at a time on this path (alternating scan streams), so a launch's duration
        # (what `achieved` is computed from) is about twice its share of the memory pipe; the loop-level
        # rate says what the GPU actually streamed.
        result["roofline"]["scans_in_flight"] = int(os.environ.get("DEWI_BENCH_SCAN_STREAMS", "2"))
        result["roofline"]["effective_GBps_per_gpu"] = round(algo_bytes * args.steps / elapsed / 1e9, 1)
        # what the process group actually was: every rank contributes (rank, device ordinal, rows held) through the
        # same backend the exchange used, so the record shows that the collective library saw N ranks
        on_dev = dist.get_backend() == "nccl"
        mine = torch.tensor([rank, device.index, n_local], dtype=torch.int64, device=device if on_dev else "cpu")
        seen = torch.empty((dist.get_world_size(), 3), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(seen.view(-1), mine)
        seen = seen.cpu().tolist()
        result["rccl"] = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                          "ranks_seen": [r[0] for r in seen], "devices": [r[1] for r in seen],
                          "rows_per_rank": [r[2] for r in seen],
                          "exchange": {"collective": "all_gather_into_tensor", "bytes_per_rank_per_query": B * c * 16,
                                       "queries_per_collective": G}}

    # ------------------------------------------------------------------ sharded result == single-GPU result
    if rank == 0 and sharded:
        if world > 1 and full is not None:
            nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(full), nat.ptr(full), total_rows, args.dim,
                                                                 nat.stream_ptr()))
            fd, fe = device_payload(torch, nat, full_cols, total_rows, device)
            single = eng.DeviceCorpus(full, fd, fe, "cosine")
        elif world == 1:
            single = corpus          # forced RCCL path on one GPU: compare with the plain search
        else:
            single = None            # weak scaling: no rank holds the whole corpus
        if single is not None:
            n_chk = min(16, n_distinct)
            bad = 0
            for j in range(n_chk):
                ri, rs = single.search_device(Q[j], k, eta, 0.0)
                torch.cuda.synchronize()
                if not (torch.equal(ri, out_ids[j]) and torch.equal(rs, out_sc[j])):
                    bad += 1
            result["sharded_parity"] = {"queries_checked": n_chk, "mismatches_vs_single_gpu": bad}
            if bad:
                print(f"SHARDED PARITY FAIL: {bad}/{n_chk} queries differ from the single-GPU search", file=sys.stderr)

    # ------------------------------------------------------------------ p50 latency of the sharded path
    # One query at a time, nothing in flight: scan + select on every rank, all-gather, merge, host sync.
    # Every rank runs the loop (the all-gather is a collective); rank 0 reports.
    if sharded and args.latency_queries > 0:
        n_lat = min(args.latency_queries, n_distinct, 300)
        barrier()
        lat = []
        for j in range(n_lat):
            t1 = time.perf_counter()
            pipe.submit(qs[j], out_records=send[0][0])
            gather_records(recv[0][: world * B * c * 4], send[0][:1].view(-1)).wait()
            eng.merge_rerank_device(recv[0][: world * B * c * 4].view(world, B, c, 4), c, k, eta, 0.0, oi[j], osc[j])
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        pipe.drain()
        # the exchange alone: one all-gather of one query's records, waited for on the host, nothing else in flight
        barrier()
        xch = []
        for j in range(min(n_lat, 100)):
            t1 = time.perf_counter()
            gather_records(recv[0][: world * B * c * 4], send[0][:1].view(-1)).wait()
            torch.cuda.synchronize()
            xch.append(time.perf_counter() - t1)
        if rank == 0 and len(xch) > 5:
            result["rccl"]["exchange"]["alone_p50_ms"] = round(float(np.percentile(np.array(xch[5:]) * 1e3, 50)), 4)
        lat = np.array(lat[5:]) * 1e3
        if rank == 0 and len(lat):
            result["p50_latency_ms"] = round(float(np.percentile(lat, 50)), 4)
            result["p99_latency_ms"] = round(float(np.percentile(lat, 99)), 4)
            result["latency_path"] = "device-resident query -> scan+select per shard -> RCCL all-gather -> merge -> host sync"

    # ------------------------------------------------------------------ p50 latency through the API
    if rank == 0 and world == 1 and not sharded and args.latency_queries > 0:
        qh = Q[: args.latency_queries, 0].cpu().numpy()
        lat = []
        for j in range(min(args.latency_queries, qh.shape[0])):
            t1 = time.perf_counter()
            corpus.search(qh[j], k, eta, 0.0)
            lat.append(time.perf_counter() - t1)
        lat = np.array(lat[5:]) * 1e3
        result["p50_latency_ms"] = round(float(np.percentile(lat, 50)), 4)
        result["p99_latency_ms"] = round(float(np.percentile(lat, 99)), 4)

    # ------------------------------------------------------------------ CPU baseline + parity gate (rank 0, N=1)
    if rank == 0 and world == 1 and n_cpu > 0:
        orc, compare_query = oracle_imports()
        from threadpoolctl import threadpool_limits
        E = host_copy_first_touched_in_parallel(corpus.emb)
        d32, e32 = dewi32.cpu().numpy(), ent32.cpu().numpy()
        nq = min(n_cpu, n_distinct)
        # GPU answers for exactly these queries (the timed run may have covered fewer of them)
        torch.cuda.set_stream(torch.cuda.default_stream()) if sharded else None
        for j in range(nq):
            corpus.search_device(Q[j], k, eta, 0.0, out_ids[j], out_sc[j])
        torch.cuda.synchronize()
        qh = Q[:nq, 0].cpu().numpy()
        gi = out_ids[:nq, 0].cpu().numpy()
        gs = out_sc[:nq, 0].cpu().numpy()
        threads, probe, avail = best_blas_threads(lambda: orc.search(E, qh[0], d32, e32, k, eta, 0.0))
        lat = []
        ref = []
        budget = time.perf_counter() + 30.0
        with threadpool_limits(limits=threads, user_api="blas"):
            for j in range(5):                                        # warm-up (5 + >= 50 timed)
                orc.search(E, qh[j % nq], d32, e32, k, eta, 0.0)
            for j in range(nq):
                t1 = time.perf_counter()
                ref.append(orc.search(E, qh[j], d32, e32, k, eta, 0.0))
                lat.append(time.perf_counter() - t1)
                if time.perf_counter() > budget and len(lat) >= 50:
                    break
        lat = np.array(lat)
        result["cpu_baseline"] = {"value": round(len(lat) / float(lat.sum()), 2), "unit": "queries/s", "cores": threads,
                                  "kind": "port", "p50_ms": round(float(np.percentile(lat, 50) * 1e3), 3),
                                  "logical_cpus_visible": avail, "host": host_facts(),
                                  "thread_probe_ms": {str(t): round(v * 1e3, 2) for t, v in probe.items()},
                                  "matrix_first_touch": "parallel (chunks of 4096 rows round-robin over a thread pool)",
                                  "sample": f"{len(lat)} single queries of the same workload (full {total_rows}x{args.dim} "
                                            f"corpus) after 5 warm-up queries, NumPy/OpenBLAS oracle at its fastest BLAS "
                                            f"thread count of the probe ladder"}
        # the reference's bare step sequence (E @ q + argpartition, backends.py:431-444) beside the full oracle
        with threadpool_limits(limits=threads, user_api="blas"):
            t1 = time.perf_counter()
            for j in range(min(20, nq)):
                s = E @ orc.prepare_query(qh[j])
                np.argpartition(s, -c)[-c:]
            bare = (time.perf_counter() - t1) / min(20, nq)
        result["cpu_baseline"]["matvec_argpartition_ms"] = round(bare * 1e3, 3)
        # parity gate: GPU results vs the oracle on the same queries
        checked = min(len(ref), 64)
        bad, near = 0, 0
        for j in range(checked):
            decisive, msg = compare_query(E, qh[j], d32, e32, k, eta, 0.0, "cosine", gi[j], gs[j], exact_gaps=False)
            near += 0 if decisive else 1
            if msg is not None:
                bad += 1
                print(f"PARITY FAIL query {j}: {msg}", file=sys.stderr)
        result["parity"] = {"queries_checked": checked, "mismatches": bad, "decisive": checked - near,
                            "near_tie_checked_by_id_set": near}
        if bad:
            print(json.dumps(result))
            raise SystemExit("parity gate failed: the bench result is invalid")
        # C1 (configs[0]) through the full add / build / search API, GPU and CPU oracle side by side
        result["c1_api"] = c1_api_leg(orc, threads)
    # ------------------------------------------------------------------ informational: the same queries through the opt-in bf16 shadow
    # NOT part of `value` (the timed region above is the plain fp32 scan over N*d*4 bytes): the one-query search of a corpus
    # that also keeps a bf16 copy (+50 % memory) — pre-selection over the copy, candidates re-scored from the fp32 rows, answers
    # checked here to be the fp32 scan's bit for bit (its own line: --batch 1 --shadow 1).
    if (rank == 0 and world == 1 and not sharded and B == 1 and not shadowed and args.shadow < 0 and args.dim in (256, 512, 768, 1024, 1536)
            and c <= 256 and n_local >= 64 * 1024):
        result["opt_in_bf16_shadow"] = shadow_leg(torch, corpus, Q, k, eta, min(64, n_distinct))
    return result


def shadow_leg(torch, corpus, Q, k, eta, n_check):
    try:
        dev = corpus.device
        want_ids = torch.empty((n_check, 1, k), dtype=torch.int64, device=dev)
        want_sc = torch.empty((n_check, 1, k), dtype=torch.float32, device=dev)
        got_ids, got_sc = torch.empty_like(want_ids), torch.empty_like(want_sc)
        for j in range(n_check):
            corpus.search_device(Q[j], k, eta, 0.0, want_ids[j], want_sc[j])
        corpus.enable_bf16_shadow(single_query=True)
        for j in range(n_check):
            corpus.search_device(Q[j], k, eta, 0.0, got_ids[j], got_sc[j])
        torch.cuda.synchronize()
        refused = int((got_ids[:, 0, 0] < 0).sum())
        ok = got_ids[:, 0, 0] >= 0
        equal = bool(torch.equal(got_ids[ok], want_ids[ok]) and torch.equal(got_sc[ok], want_sc[ok]))
        steps, n_q = 500, int(Q.shape[0])
        for i in range(50):
            corpus.search_device(Q[i % n_q], k, eta, 0.0, got_ids[i % n_check], got_sc[i % n_check])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            corpus.search_device(Q[i % n_q], k, eta, 0.0, got_ids[i % n_check], got_sc[i % n_check])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"queries_per_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 5), "steps": steps,
                "answers_bit_equal_to_the_fp32_scan": equal, "queries_compared": int(ok.sum()), "refused": refused,
                "extra_hbm_bytes": int(corpus.shadow.numel() * 2),
                "note": "informational, not `value`: enable_bf16_shadow(single_query=True) — bf16 copy pre-selects, fp32 rows re-score"}
    except Exception as e:  # noqa: BLE001  (an extra leg must never cost the headline line)
        return {"error": repr(e)[:300]}
    finally:
        corpus.shadow = None
        corpus.shadow_min_batch = 2


def c1_api_leg(orc, threads):
    """BASELINE configs[0]: 10 K docs, d=768, k=10 through the public API (DewiIndex add_batch -> build ->
    search), with the oracle's per-query loop on the same data as the CPU side."""
    from threadpoolctl import threadpool_limits
    from dewi.index import DewiIndex
    from dewi.types import payloads_from_columns
    n, d, k, eta = 10_000, 768, 10, 0.3
    raw = orc.synth_corpus(n, d, seed=42)
    cols = orc.synth_payload_columns(n, seed=42)
    Q = orc.synth_queries(64, d, seed=7)
    t0 = time.perf_counter()
    index = DewiIndex(dim=d, use_ann=False, rerank_eta=eta)
    index.add_batch([f"doc_{i:08d}" for i in range(n)], raw, payloads_from_columns(cols))
    index.build()
    build_s = time.perf_counter() - t0
    for q in Q[:5]:
        index.search(q, k=k)
    lat = []
    for q in Q:
        t1 = time.perf_counter()
        index.search(q, k=k)
        lat.append(time.perf_counter() - t1)
    E = orc.build_matrix(raw)
    d32, e32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    cl = []
    with threadpool_limits(limits=threads, user_api="blas"):
        for q in Q:
            t1 = time.perf_counter()
            orc.search(E, q, d32, e32, k, eta, 0.0)
            cl.append(time.perf_counter() - t1)
    return {"workload": "10K docs x d=768, k=10, eta=0.3 through DewiIndex.add_batch/build/search (BASELINE configs[0])",
            "gpu_build_s": round(build_s, 4), "gpu_search_p50_ms": round(float(np.percentile(lat, 50)) * 1e3, 4),
            "gpu_queries_per_s": round(len(lat) / sum(lat), 1),
            "cpu_oracle_search_p50_ms": round(float(np.percentile(cl, 50)) * 1e3, 4),
            "cpu_oracle_queries_per_s": round(len(cl) / sum(cl), 1)}


# ======================================================================================================
# c3: 1M x 768 bf16, 256 queries per step, k=100 — the batched matrix-core path
# ======================================================================================================
def run_c3(args, torch, eng, nat, device):
    n, dim, B, k, eta = args.docs, args.dim, args.batch, args.k, args.eta
    emb, cols = make_corpus(torch, n, dim, 42, device)
    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(emb), nat.ptr(emb), n, dim, nat.stream_ptr()))
    dewi32, ent32 = device_payload(torch, nat, cols, n, device)
    cb = eng.DeviceCorpus(emb, dewi32, ent32, "cosine").to_bf16()
    del emb
    qg = torch.Generator(device=device)
    qg.manual_seed(7)
    n_batches = 8
    Q = torch.randn((n_batches, B, dim), generator=qg, device=device, dtype=torch.float32)
    out_ids = torch.empty((n_batches, B, k), dtype=torch.int64, device=device)
    out_sc = torch.empty((n_batches, B, k), dtype=torch.float32, device=device)

    def run(first, count):
        for i in range(first, first + count):
            j = i % n_batches
            cb.search_device(Q[j], k, eta, 0.0, out_ids[j], out_sc[j])

    eng.timing(1)          # fills the library's event pool outside the timed region (see run_c2)
    run(0, args.warmup)
    torch.cuda.synchronize()
    eng.timing_read()
    eng.timing(4)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    region_ms, region_launches = eng.timing_read()
    eng.timing(1)
    run(0, MIN_ROOFLINE_LAUNCHES)
    torch.cuda.synchronize()
    leg_ms, leg_launches = eng.timing_read()
    eng.timing(False)
    tot = region_launches + leg_launches
    kern_ms = (region_ms * region_launches + leg_ms * leg_launches) / tot if tot else 0.0
    refused = int((out_ids[:, :, 0] < 0).sum().item())
    ms_per_step = elapsed / args.steps * 1e3
    algo_bytes = n * dim * 2 + B * dim * 2
    flops = 2.0 * B * n * dim
    hbm = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    tf = flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    kernel = "mfma_scan_bf16_s16<48,false>"
    traffic, traffic_note = recorded_traffic(f"{n}x{dim}x2xB{B}", kernel)
    result = {
        "metric": "queries/sec, 1M×768 bf16 corpus, query batch=256, k=100, η=0.3 (BASELINE.json configs[2])",
        "value": round(args.steps * B / elapsed, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{n} docs x d={dim} bf16, query batch={B}, k={k}, eta={eta}, batched matrix-core kNN "
                               f"+ DEWI re-rank (BASELINE.json configs[2])",
                   "docs": n, "dim": dim, "k": k, "eta": eta, "batch": B, "candidates": min(2 * k, n),
                   "parallelism": "single GPU", "refused_queries": refused},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(hbm / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": algo_bytes, "mean_kernel_ms": round(kern_ms, 5), "launches_timed": tot,
                     "mfma": {"achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                              "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "flops_per_launch": flops},
                     "whole_step": {"ms": round(ms_per_step, 5),
                                    "hbm_frac": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "mfma_frac": round(flops / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}},
    }
    if args.cpu_queries > 0:
        orc, compare_query = oracle_imports()
        from threadpoolctl import threadpool_limits
        Eb = host_copy_first_touched_in_parallel(cb.emb.float())
        d32, e32 = dewi32.cpu().numpy(), ent32.cpu().numpy()
        nq = min(max(50, args.cpu_queries), B)
        Qp = eng.prepare_queries_bf16(Q[0]).float().cpu().numpy()      # the device's own prepared queries
        want = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q[0, :nq].cpu().numpy()])
        prep_equal = float(np.mean(Qp[:nq] == want))
        gi, gs = out_ids[0].cpu().numpy(), out_sc[0].cpu().numpy()
        threads, probe, avail = best_blas_threads(lambda: orc.search_prepared(Eb, Qp[0], d32, e32, k, eta, 0.0))
        lat = []
        with threadpool_limits(limits=threads, user_api="blas"):
            for j in range(nq):
                t1 = time.perf_counter()
                orc.search_prepared(Eb, Qp[j], d32, e32, k, eta, 0.0)
                lat.append(time.perf_counter() - t1)
        lat = np.array(lat)
        result["cpu_baseline"] = {"value": round(len(lat) / float(lat.sum()), 2), "unit": "queries/s", "cores": threads,
                                  "kind": "port", "p50_ms": round(float(np.percentile(lat, 50) * 1e3), 3),
                                  "logical_cpus_visible": avail,
                                  "sample": f"{len(lat)} queries of one batch, one at a time (the reference has no batch "
                                            f"API), bf16-rounded inputs, NumPy/OpenBLAS oracle"}
        bad = near = 0
        checked = min(nq, 32)
        for j in range(checked):
            decisive, msg = compare_query(Eb, Qp[j], d32, e32, k, eta, 0.0, "cosine", gi[j], gs[j], exact_gaps=False,
                                          gap=1e-6, prepared=True)
            near += 0 if decisive else 1
            if msg is not None:
                bad += 1
                print(f"PARITY FAIL query {j}: {msg}", file=sys.stderr)
        result["parity"] = {"queries_checked": checked, "mismatches": bad, "decisive": checked - near,
                            "near_tie_checked_by_id_set": near, "prepared_query_elements_bit_equal": round(prep_equal, 5)}
        if bad:
            print(json.dumps(result))
            raise SystemExit("parity gate failed: the bench result is invalid")
    return result


# ======================================================================================================
# c4: configs[3] replayed on one GPU — 8 resident 1M-row shards, scan each, merge
# ======================================================================================================
def run_c4(args, torch, eng, nat, device):
    S, n, dim, B, k, eta = args.emulate_shards, args.docs, args.dim, args.batch, args.k, args.eta
    total = S * n
    c = min(2 * k, total)
    big = torch.empty((total, dim), dtype=torch.float32, device=device)
    dewi_all = torch.empty(total, dtype=torch.float32, device=device)
    ent_all = torch.empty(total, dtype=torch.float32, device=device)
    shards = []
    for s in range(S):                                  # seeds 42..49 as SURVEY §8(d) prescribes for C4
        emb, cols = make_corpus(torch, n, dim, 42 + s, device)
        big[s * n:(s + 1) * n] = emb
        del emb
        d32, e32 = device_payload(torch, nat, cols, n, device)
        dewi_all[s * n:(s + 1) * n] = d32
        ent_all[s * n:(s + 1) * n] = e32
    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(big), nat.ptr(big), total, dim, nat.stream_ptr()))
    for s in range(S):
        shards.append(eng.DeviceCorpus(big[s * n:(s + 1) * n], dewi_all[s * n:(s + 1) * n], ent_all[s * n:(s + 1) * n],
                                       "cosine", id_offset=s * n))
    whole = eng.DeviceCorpus(big, dewi_all, ent_all, "cosine")
    qg = torch.Generator(device=device)
    qg.manual_seed(7)
    n_distinct = 256
    Q = torch.randn((n_distinct, B, dim), generator=qg, device=device, dtype=torch.float32)
    out_ids = torch.empty((n_distinct, B, k), dtype=torch.int64, device=device)
    out_sc = torch.empty((n_distinct, B, k), dtype=torch.float32, device=device)
    recs = torch.empty((S, B, c, 4), dtype=torch.int32, device=device)

    def run(first, count):
        for i in range(first, first + count):
            j = i % n_distinct
            for s in range(S):
                shards[s].candidates_device(Q[j], c, out=recs[s])
            eng.merge_rerank_device(recs, c, k, eta, 0.0, out_ids[j], out_sc[j])

    eng.timing(1)          # fills the library's event pool outside the timed region (see run_c2)
    run(0, args.warmup)
    torch.cuda.synchronize()
    eng.timing_read()
    eng.timing(8)
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = eng.timing_read()
    eng.timing(1)
    if launches < MIN_ROOFLINE_LAUNCHES:
        run(0, (MIN_ROOFLINE_LAUNCHES - launches + S - 1) // S)
        torch.cuda.synchronize()
        ms2, l2 = eng.timing_read()
        kern_ms, launches = (kern_ms * launches + ms2 * l2) / (launches + l2), launches + l2
    eng.timing(False)
    # the emulated exchange must equal ONE 8M-row search bit for bit
    bad = 0
    n_chk = 16
    for j in range(n_chk):
        ri, rs = whole.search_device(Q[j], k, eta, 0.0)
        bad += 0 if (torch.equal(ri, out_ids[j]) and torch.equal(rs, out_sc[j])) else 1
    torch.cuda.synchronize()
    algo_bytes = n * dim * 4 + B * dim * 4
    hbm = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3
    c4_traffic, c4_note = recorded_traffic(f"{n}x{dim}x4xB{B}", "scan_rows_f32")
    result = {
        "metric": "queries/sec, 8M×768 fp32 corpus as 8 doc-id shards of 1M, k=10, η=0.3 (BASELINE.json configs[3]) "
                  "REPLAYED ON ONE GPU: every shard scanned in turn, records merged; no wire",
        "value": round(args.steps * B / elapsed, 2), "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{S} shards x {n} docs x d={dim} fp32 resident on one GPU ({total * dim * 4 / 1e9:.1f} GB), "
                               f"query batch={B}, k={k}, eta={eta}: dewi_knn_candidates per shard + dewi_merge_rerank",
                   "docs": total, "dim": dim, "k": k, "eta": eta, "batch": B, "candidates": c, "shards": S,
                   "parallelism": f"{S} shards emulated on a single GPU (multi-GPU: unmeasured here)",
                   "ideal_8gpu_ms_per_step": round(ms_per_step / S, 5)},
        "roofline": {"bound": "hbm", "kernel": "scan_rows_f32", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4), "traffic": c4_traffic,
                     "traffic_source": c4_note + " (the same kernel over one 1M-row shard: the C2 record)",
                     "algorithmic_bytes_per_launch": algo_bytes, "mean_kernel_ms": round(kern_ms, 5),
                     "launches_timed": launches},
        "sharded_parity": {"queries_checked": n_chk, "mismatches_vs_single_8M_search": bad},
    }
    if args.cpu_queries > 0:
        orc, compare_query = oracle_imports()
        E = big.cpu().numpy()
        d32, e32 = dewi_all.cpu().numpy(), ent_all.cpu().numpy()
        nq = 8
        qh = Q[:nq, 0].cpu().numpy()
        gi, gs = out_ids[:nq, 0].cpu().numpy(), out_sc[:nq, 0].cpu().numpy()
        lat = []
        msgs = 0
        for j in range(nq):
            t1 = time.perf_counter()
            orc.search(E, qh[j], d32, e32, k, eta, 0.0)
            lat.append(time.perf_counter() - t1)
            decisive, msg = compare_query(E, qh[j], d32, e32, k, eta, 0.0, "cosine", gi[j], gs[j], exact_gaps=False)
            if msg is not None:
                msgs += 1
                print(f"PARITY FAIL query {j}: {msg}", file=sys.stderr)
        result["cpu_baseline"] = {"value": round(len(lat) / sum(lat), 3), "unit": "queries/s", "cores": os.cpu_count(),
                                  "kind": "port", "sample": f"{nq} single queries over the whole {total}-row corpus"}
        result["parity"] = {"queries_checked": nq, "mismatches": msgs}
        if msgs or bad:
            print(json.dumps(result))
            raise SystemExit("parity gate failed: the bench result is invalid")
    return result


# ======================================================================================================
# c5: 1M documents, d=512 — robust fit + score of the 7 signals, I_hat row-cosine
# ======================================================================================================
def run_c5(args, torch, nat, device):
    """The on-GPU scorer part of configs[4] THROUGH THE PYTHON LAYER, device tensors in and out (the reference's caller:
    pipelines.py:180-223): I_hat = signals.cross_modal_similarity(text, image) written into its row of the [7][N] signal
    table, DewiScorer.fit_stats_columns on the table's rows (read in place), score_batch_device -> the fp32 dewi column an
    index ingests.  Nothing N long visits the host, nothing is synchronised inside a step.  The same three kernels
    driven by bare ctypes calls are timed beside it (`c_abi_loop`): the Python layer must cost nothing measurable."""
    import ctypes
    from dewi import signals
    from dewi.scorer import DewiScorer
    from dewi.types import SIGNAL_FIELDS
    lib = nat.load_library()
    n, dim = args.docs, args.dim
    rs = np.random.RandomState(1042)
    sig = np.stack([rs.gamma(2, 0.5, n), rs.gamma(2, 0.5, n) * 1.5, rs.gamma(2, 0.3, n), rs.gamma(2, 0.3, n) * 1.5,
                    rs.beta(2, 2, n), rs.beta(1, 5, n), rs.beta(1, 10, n)]).astype(np.float32)
    S = torch.from_numpy(sig).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(42)
    A = torch.randn((n, dim), generator=g, device=device)
    g.manual_seed(43)
    Bm = torch.randn((n, dim), generator=g, device=device)
    med = torch.empty(7, dtype=torch.float32, device=device)
    mad = torch.empty(7, dtype=torch.float32, device=device)
    wsb = int(lib.dewi_robust_fit_workspace_bytes(7))
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    out64 = torch.empty(n, dtype=torch.float64, device=device)
    out32 = torch.empty(n, dtype=torch.float32, device=device)
    ihat = torch.empty(n, dtype=torch.float32, device=device)
    arr7, arr5 = ctypes.c_double * 7, ctypes.c_double * 5
    st = nat.stream_ptr()

    # ---- the Python API on device tensors
    scorer = DewiScorer()
    cols = {key: S[j] for j, key in enumerate(SIGNAL_FIELDS)}            # views of the table: read in place
    ihat_row = torch.empty(n, dtype=torch.float32, device=device)        # (the synthetic I_hat column of `sig` is what is
    api_out = {}                                                         # scored, so that the oracle check below holds)

    def api_step():
        signals.cross_modal_similarity(A, Bm, return_device=True, out=ihat_row)
        scorer.fit_stats_columns(cols)
        api_out["f64"], api_out["dewi32"] = scorer.score_batch_device(cols)

    # ---- the same kernels through bare ctypes calls
    def fit():
        nat.check(lib.dewi_robust_fit_f32(nat.ptr(S), n, n, 7, nat.ptr(med), nat.ptr(mad), nat.ptr(ws), wsb, st))

    fit()
    torch.cuda.synchronize()
    mh = med.cpu().numpy().astype(np.float64)
    dh = np.array([float(x) or 1e-8 for x in mad.cpu().numpy().astype(np.float64)])

    def score():
        nat.check(lib.dewi_score_f64(nat.ptr(S), 0, n, n, arr7(*mh), arr7(*dh), arr5(1, 1, 1, 1, 1), 3.0, 0, nat.ptr(out64),
                                     nat.ptr(out32), st))

    def cosine():
        nat.check(lib.dewi_row_cosine_f32(nat.ptr(A), nat.ptr(Bm), nat.ptr(ihat), n, dim, st))

    def raw_step():
        cosine()
        fit()
        score()

    def loop(step, count):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(count):
            step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        api_step()
    for _ in range(args.warmup):
        raw_step()
    elapsed = loop(api_step, args.steps)                      # THE timed region: the Python API
    raw_runs = [loop(raw_step, args.steps) for _ in range(3)]
    api_runs = [elapsed] + [loop(api_step, args.steps) for _ in range(2)]
    raw_ms = float(np.median(raw_runs)) / args.steps * 1e3
    api_ms = float(np.median(api_runs)) / args.steps * 1e3
    same = bool(torch.equal(api_out["f64"], out64) and torch.equal(api_out["dewi32"], out32) and torch.equal(ihat_row, ihat))
    stats_same = all(scorer.stats.medians[k] == float(mh[j]) and scorer.stats.mads[k] == float(dh[j])
                     for j, k in enumerate(SIGNAL_FIELDS))
    if api_ms > 1.05 * raw_ms:
        print(f"WARNING: the Python API loop ({api_ms:.4f} ms per step) is more than 5 % behind the bare C-ABI loop "
              f"({raw_ms:.4f} ms)", file=sys.stderr)

    def timed(fn, reps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:                      # kernels go on torch's current stream, so torch events see them
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    cos_ms = timed(cosine, MIN_ROOFLINE_LAUNCHES)
    fit_ms = timed(fit, MIN_ROOFLINE_LAUNCHES)
    score_ms = timed(score, MIN_ROOFLINE_LAUNCHES)
    cos_bytes = 2 * n * dim * 4 + n * 4
    fit_bytes = 2 * 7 * n * 4                       # one median pass + one MAD pass
    score_bytes = 7 * n * 4 + n * 4
    ms_per_step = elapsed / args.steps * 1e3
    cos_traffic, cos_note = recorded_traffic(f"rowcos_{n}x{dim}", "row_cosine_512_kernel<2>")
    fit_t = [recorded_traffic(f"fit_{ph}_7x{n}", f"fit_fast_kernel<{'true' if ph == 'mad' else 'false'}>")[0] for ph in ("med", "mad")]
    fit_note = recorded_traffic(f"fit_med_7x{n}", "fit_fast_kernel<false>")[1]
    result = {
        "metric": "documents/sec through the on-GPU scorer part of BASELINE.json configs[4]: I_hat row-cosine (d=512) + "
                  "robust fit (7 signals) + DEWI score, 1M documents",
        "value": round(args.steps * n / elapsed, 1), "unit": "documents/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32 (fit, cosine) / f64 (score)", "data": "synthetic",
        "config": {"workload": f"{n} documents: row-cosine of two {n}x{dim} fp32 matrices, exact median/MAD of 7 fp32 "
                               f"signal columns, float64 DEWI score (BASELINE.json configs[4], scorer part)",
                   "docs": n, "dim": dim, "signals": 7,
                   "driven_by": "dewi.signals.cross_modal_similarity + DewiScorer.fit_stats_columns + score_batch_device "
                                "on CUDA tensors (device-resident, no synchronisation inside a step)"},
        "python_api_vs_c_abi": {"api_ms_per_step": round(api_ms, 5), "c_abi_ms_per_step": round(raw_ms, 5),
                                "ratio": round(api_ms / raw_ms, 4), "results_bit_equal": same and stats_same},
        "roofline": {"bound": "hbm", "kernel": "row_cosine_512_kernel", "achieved": round(cos_bytes / (cos_ms * 1e-3) / 1e9, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(cos_bytes / (cos_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": cos_traffic, "traffic_source": cos_note,
                     "algorithmic_bytes_per_launch": cos_bytes, "mean_kernel_ms": round(cos_ms, 5),
                     "launches_timed": MIN_ROOFLINE_LAUNCHES,
                     "robust_fit": {"ms": round(fit_ms, 5), "algorithmic_bytes": fit_bytes,
                                    "frac": round(fit_bytes / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "traffic": (fit_t[0] + fit_t[1]) if None not in fit_t else None,
                                    "traffic_source": "fit_fast_kernel<false> + <true>, " + fit_note},
                     "score": {"ms": round(score_ms, 5), "algorithmic_bytes": score_bytes,
                               "frac": round(score_bytes / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
    }
    if args.cpu_queries > 0:
        orc, _ = oracle_imports()
        cols = {key: sig[j] for j, key in enumerate(orc.SIGNAL_KEYS)}
        t1 = time.perf_counter()
        m_ref, d_ref = orc.robust_fit(cols)
        fit_cpu = time.perf_counter() - t1
        t1 = time.perf_counter()
        ref = orc.score({key: v.astype(np.float64) for key, v in cols.items()}, m_ref, d_ref)
        score_cpu = time.perf_counter() - t1
        t1 = time.perf_counter()
        ncos = 100_000
        a, b = A[:ncos].cpu(), Bm[:ncos].cpu()
        t1 = time.perf_counter()
        cref = torch.nn.functional.cosine_similarity(a, b).numpy()
        cos_cpu = (time.perf_counter() - t1) * (n / ncos)
        ok_fit = all(float(mh[j]) == m_ref[key] and float(dh[j]) == d_ref[key] for j, key in enumerate(orc.SIGNAL_KEYS))
        rel = float(np.max(np.abs(out64.cpu().numpy() - ref) / ref))
        cerr = float(np.max(np.abs(ihat[:ncos].cpu().numpy() - cref)))
        result["cpu_baseline"] = {"value": round(n / (fit_cpu + score_cpu + cos_cpu), 1), "unit": "documents/s",
                                  "cores": os.cpu_count(), "kind": "port",
                                  "sample": f"NumPy oracle: fit {fit_cpu:.3f} s + score {score_cpu:.3f} s on all {n} "
                                            f"documents; torch CPU cosine on {ncos} rows scaled to {n}"}
        result["parity"] = {"medians_mads_bit_exact": bool(ok_fit), "score_max_rel_err": rel, "cosine_max_abs_err": cerr}
        if not ok_fit or rel > 1e-15 or cerr > 2e-6:
            print(json.dumps(result))
            raise SystemExit("parity gate failed: the bench result is invalid")
    return result


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): start the N ranks as a CHILD
    `python -m torch.distributed.run` of this very command line and hand its exit code back.  Nothing in this process
    has touched the GPU (torch is not even imported yet), and nothing is exec'ed: the child inherits stdout, so rank 0's
    JSON line reaches whoever reads ours.  DEWI_BENCH_LAUNCH_DRYRUN=1 prints the command instead of running it."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    if os.environ.get("DEWI_BENCH_LAUNCH_DRYRUN", "0") == "1":
        print(json.dumps({"launcher": cmd}))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    print(f"bench.py: --gpus {n_ranks} without a launcher: starting {n_ranks} ranks with torch.distributed.run "
          f"(127.0.0.1:{port})", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): running at {world}", file=sys.stderr)
        args.gpus = world
    if world > 1 and args.config != "c2":
        raise SystemExit("--config c3/c4/c5 are single-GPU harness legs; the multi-GPU run is the default config")
    # DEWI_BENCH_BACKEND=gloo DEWI_BENCH_DEVICE=0: REHEARSAL of the multi-rank loop (slicing, grouping, merge, parity
    # and latency legs) with several ranks sharing one GPU and records staged through the host — its numbers mean nothing
    backend = os.environ.get("DEWI_BENCH_BACKEND", "nccl")
    if "DEWI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DEWI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=device)
            else:
                dist.init_process_group(backend=backend)
            dist.barrier()                                   # brings the communicator (and RCCL's banner) up here

    from dewi import _engine as eng
    from dewi import _native as nat
    nat.load_library()
    eng.tuning(args.scan_blocks, args.rows_per_iter, args.nontemporal)

    if args.config == "c2":
        result = run_c2(args, torch, dist, eng, nat, rank, world, device)
    elif args.config == "c3":
        result = run_c3(args, torch, eng, nat, device)
    elif args.config == "c4":
        result = run_c4(args, torch, eng, nat, device)
    else:
        result = run_c5(args, torch, nat, device)
    result["sources_sha256"] = sources_digest()[:16]
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
