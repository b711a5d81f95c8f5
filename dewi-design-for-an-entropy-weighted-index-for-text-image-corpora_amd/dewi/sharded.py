"""Doc-id sharded search across the GPUs of one node.

The reference is single-process; this is the MI355X-native addition.  One process per GPU
(``torch.distributed``; backend ``nccl`` is RCCL over xGMI).  Rank r owns the contiguous row range
``[offset_r, offset_r + n_r)`` of the corpus plus the matching payload columns.  Per query batch:

  1. every rank scans its shard and keeps its best ``c = min(2k, N_global)`` rows by similarity as
     16-byte records (sim, dewi32, ent32, global id), sorted — ``dewi_knn_candidates``;
  2. ONE all-gather of ``B x c`` records per rank (k=10, B=1: 320 B — latency-bound, so a single
     small collective, not a reduction tree);
  3. every rank selects the global top-c by (sim desc, id asc), applies the eta blend and takes the
     top-k — ``dewi_merge_rerank`` — so all ranks hold the identical answer.

The similarity cut is global and happens before the blend, exactly as in the single-device path,
so the result is independent of the sharding.

``scan_fn`` / ``merge_fn`` exist so that the exchange logic (shard arithmetic, record layout,
gather order, padding of short shards) can be exercised on CPU-only hosts with ``gloo``; the
defaults are the HIP kernels and there is no CPU implementation in this package.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

RECORD_WORDS = 4  # sim, dewi, ent (fp32 bit patterns) + id, as int32 words


def shard_bounds(n_total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced row ranges with EVEN boundaries: shard r = [cut(r), cut(r+1)), cut(r) = n*r//world
    rounded to the nearest even row (the last shard ends at n; sizes differ by at most 3).  Why even: the one-query kernel of a bf16 corpus takes
    1536-byte rows (dim 768) in PAIRS, and a row's fp32 sum is accumulated in a different lane order for the first
    and the second row of a pair; a shard that starts on an odd row flips every row's place in its pair, and its
    scores then agree with the whole-corpus search to summation noise (~1e-7) instead of bit for bit.  With even
    boundaries "sharded == single device, bit for bit" holds for every element type and kernel."""
    cuts = [((n_total * r) // world + 1) & ~1 for r in range(world)] + [n_total]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class ShardedSearcher:
    """Search over a corpus sharded by doc id; one instance per rank."""

    def __init__(self, local, n_local: int, group=None, scan_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None, device=None, always_collective: bool = False):
        """``local``: this rank's ``DeviceCorpus`` (or any object, when ``scan_fn`` is given).

        ``always_collective``: run the all-gather even in a group of one rank (by default a lone rank
        skips it) — the single-GPU rehearsal of the RCCL path used by the tests and by ``bench.py``.

        ``scan_fn(queries, c) -> int32 tensor [B, c, 4]`` of records sorted by (sim desc, id asc)
        with ids already global and ``id = -1`` padding; ``merge_fn(lists[world, B, c, 4], c, k,
        eta, entropy_pref) -> (ids int64 [B, k], scores fp32 [B, k])``.
        """
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.always_collective = bool(always_collective) and dist.is_initialized()
        self.device = device if device is not None else getattr(local, "device", torch.device("cpu"))
        # exclusive prefix sum of the shard sizes = id offsets; all ranks learn every size
        sizes = [int(n_local)]
        if self.world > 1:
            gathered: List[Optional[int]] = [None] * self.world
            dist.all_gather_object(gathered, int(n_local), group=group)
            sizes = [int(s) for s in gathered]
        self.sizes = sizes
        self.n_total = int(sum(sizes))
        self.id_offset = int(sum(sizes[: self.rank]))
        have = getattr(local, "id_offset", self.id_offset)
        if have != self.id_offset:
            raise ValueError(f"rank {self.rank}: shard id_offset {have} != prefix sum of shard sizes {self.id_offset}")
        if scan_fn is None:
            scan_fn = lambda q, c: local.candidates_device(local.stage_queries(q), c)  # noqa: E731
        if merge_fn is None:
            from ._engine import merge_rerank_device
            merge_fn = merge_rerank_device
        self._scan, self._merge = scan_fn, merge_fn

    def n_candidates(self, k: int) -> int:
        return min(2 * int(k), self.n_total)          # reference backends.py:439, over the WHOLE corpus

    # ------------------------------------------------------------------ exchange
    def exchange(self, recs):
        """all-gather [B, c, 4] int32 records -> [world, B, c, 4] on every rank."""
        torch, dist = self._torch, self._dist
        if self.world == 1 and not self.always_collective:
            return recs.unsqueeze(0)
        if self.backend == "nccl":
            out = torch.empty((self.world,) + tuple(recs.shape), dtype=recs.dtype, device=recs.device)
            dist.all_gather_into_tensor(out.view(-1), recs.contiguous().view(-1), group=self.group)
            return out
        # gloo (CPU rehearsal, or several ranks sharing one GPU): stage through host memory
        host = recs.detach().cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(self.world)]
        dist.all_gather(parts, host, group=self.group)
        return torch.stack(parts).to(recs.device)

    # ------------------------------------------------------------------ search
    def search(self, queries, k: int = 10, eta: float = 0.5, entropy_pref: float = 0.0):
        """[B, dim] queries (same on every rank) -> (global ids int64 [B, k], scores fp32 [B, k])."""
        k = int(k)
        if k <= 0:
            b = 1 if np.ndim(queries) == 1 else len(queries)
            return np.empty((b, 0), np.int64), np.empty((b, 0), np.float32)
        if k > self.n_total:
            raise ValueError(f"kth(={self.n_total - k}) out of bounds ({self.n_total})")   # as NumPy in the reference
        c = self.n_candidates(k)
        to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)  # noqa: E731

        def one_pass(q):
            lists = self.exchange(self._scan(q, c))
            ids, scores = self._merge(lists, c, k, float(eta), float(entropy_pref))
            return to_np(ids), to_np(scores)

        # (a query a shard's matrix-core pass refuses is repaired inside that shard's dewi_knn_candidates / dewi_knn_finish:
        # every record that reaches the exchange is real or padding, ABI 5)
        return one_pass(queries)


def build_local_shard(rows: np.ndarray, dewi: Sequence[float], ht_mean: Sequence[float], hi_mean: Sequence[float],
                      rank: int, world: int, space: str = "cosine", device: Optional[str] = None):
    """Slice the full host arrays to this rank's range and put the slice on the GPU."""
    from ._engine import DeviceCorpus
    lo, hi = shard_bounds(len(rows), world)[rank]
    return DeviceCorpus.from_host(rows[lo:hi], np.asarray(dewi)[lo:hi], np.asarray(ht_mean)[lo:hi],
                                  np.asarray(hi_mean)[lo:hi], space, device=device, id_offset=lo)


class HipFitSteps:
    """The HIP side of the sharded fit: this rank's ``[n_signals][n_local]`` fp32 device table plus
    the workspace of ``dewi_robust_fit_*`` (include/dewi_hip.h, "A6 over doc-id shards")."""

    def __init__(self, table):
        import torch
        from . import _native as nat
        self._nat, self._torch = nat, torch
        self.lib = nat.load_library()
        if table.dtype != torch.float32 or table.dim() != 2 or not table.is_cuda:
            raise ValueError("table must be a CUDA fp32 tensor [n_signals][n_local]")
        self.table = table.contiguous()
        self.n_signals, self.n_local = int(table.shape[0]), int(table.shape[1])
        self.ws_bytes = int(self.lib.dewi_robust_fit_workspace_bytes(self.n_signals))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=table.device)
        self.med = torch.zeros(self.n_signals, dtype=torch.float32, device=table.device)
        self.mad = torch.zeros(self.n_signals, dtype=torch.float32, device=table.device)

    def begin(self):
        nat = self._nat
        nat.check(self.lib.dewi_robust_fit_begin(self.n_signals, nat.ptr(self.ws), self.ws_bytes, nat.stream_ptr()))

    def hist(self, phase: int, pass_: int):
        nat = self._nat
        ld = max(self.n_local, 1)
        nat.check(self.lib.dewi_robust_fit_hist_f32(nat.ptr(self.table) if self.n_local else None, self.n_local, ld,
                                                    self.n_signals, phase, pass_, nat.ptr(self.med), nat.ptr(self.ws),
                                                    self.ws_bytes, nat.stream_ptr()))

    def regions(self, phase: int, pass_: int):
        """int32 views of the workspace regions the caller must sum over ranks."""
        import ctypes
        out = []
        for which in ((0, 1) if pass_ == 0 else (0,)):
            off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
            self._nat.check(self.lib.dewi_robust_fit_region(self.n_signals, phase, pass_, which, ctypes.byref(off),
                                                            ctypes.byref(cnt)))
            out.append(self.ws[off.value: off.value + 4 * cnt.value].view(self._torch.int32))
        return out

    def pick(self, n_total: int, phase: int, pass_: int):
        nat = self._nat
        nat.check(self.lib.dewi_robust_fit_pick(n_total, self.n_signals, phase, pass_, nat.ptr(self.ws), self.ws_bytes,
                                                nat.stream_ptr()))

    def finish(self, n_total: int, phase: int):
        nat = self._nat
        out = self.med if phase == 0 else self.mad
        nat.check(self.lib.dewi_robust_fit_finish(n_total, self.n_signals, phase, nat.ptr(self.ws), self.ws_bytes,
                                                  nat.ptr(out), nat.stream_ptr()))
        return out


class ShardedRobustFit:
    """Exact global median and MAD of signal columns whose rows are split across ranks
    (reference scorer.py:18-26 over the union of the shards).

    The three-pass radix select is order-independent, so only histograms cross the wire: per
    (phase, pass) every rank histograms its rows, the ``2*n_signals x 2048`` u32 counts are summed
    with one all-reduce (114 KB at 7 signals: latency-bound), and every rank picks the bin with the
    global row count.  6 histogram all-reduces + 2 NaN-count all-reduces per fit; the result is
    bit-identical on every rank and to the single-device fit of the concatenated rows.

    ``steps`` (default ``HipFitSteps(table)``) is injectable so that the orchestration can be
    rehearsed with ``gloo`` on CPU-only hosts; there is no CPU implementation in this package.
    """

    def __init__(self, steps, n_local: int, group=None, always_collective: bool = False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.steps, self.group = steps, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # run the all-reduces even in a group of one rank (single-GPU rehearsal of the RCCL path)
        self.always_collective = bool(always_collective) and dist.is_initialized()
        n = torch.tensor([int(n_local)], dtype=torch.int64)
        if self.world > 1 or self.always_collective:
            if self.backend == "nccl":
                n = n.cuda()
            dist.all_reduce(n, group=group)
        self.n_total = int(n.item())
        if self.n_total <= 0:
            raise IndexError("cannot fit robust statistics on an empty table")
        if self.n_total >= 2 ** 31:
            raise NotImplementedError("sharded fit sums int32 histogram counts: n_total must be below 2^31")

    def _sum_over_ranks(self, t):
        if self.world == 1 and not self.always_collective:
            return
        dist = self._dist
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, group=self.group)
            return
        host = t.detach().cpu()                      # gloo with device tensors (ranks sharing one GPU)
        dist.all_reduce(host, group=self.group)
        t.copy_(host)

    def fit(self):
        """-> (medians, MADs): fp32 arrays [n_signals], identical on every rank."""
        st = self.steps
        st.begin()
        out = []
        for phase in (0, 1):
            for pass_ in (0, 1, 2):
                st.hist(phase, pass_)
                for region in st.regions(phase, pass_):
                    self._sum_over_ranks(region)
                st.pick(self.n_total, phase, pass_)
            res = st.finish(self.n_total, phase)
            out.append(res.detach().cpu().numpy().copy() if hasattr(res, "detach") else np.array(res, dtype=np.float32))
        return out[0], out[1]
