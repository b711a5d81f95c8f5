"""Device-resident corpus + thin wrappers over the C ABI.

``DeviceCorpus`` owns what the search path needs in HBM — the N x d embedding matrix
(row-major, already normalised for cosine) and the two fp32 payload columns the re-rank
reads — and turns a query batch into (row ids, adjusted scores) with exactly two kernel
launches (scan, select/re-rank).  PyTorch is used only for device memory, streams and
host<->device copies.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from . import _native as nat

ArrayLike = Union[np.ndarray, "torch.Tensor"]  # noqa: F821


def _torch():
    import torch
    return torch


class DeviceCorpus:
    """Embedding block + payload columns of one doc-id shard, resident on one GPU."""

    def __init__(self, emb, dewi32, ent32, space: str = "cosine", id_offset: int = 0):
        torch = _torch()
        if space not in nat.SPACE_CODES:
            raise ValueError(f"unknown space {space!r}")
        assert emb.is_cuda and emb.dim() == 2 and emb.is_contiguous()
        assert emb.dtype in (torch.float32, torch.bfloat16)
        self.emb = emb
        self.dewi32 = dewi32.contiguous()
        self.ent32 = ent32.contiguous()
        assert self.dewi32.dtype == torch.float32 and self.ent32.dtype == torch.float32
        assert self.dewi32.numel() == emb.shape[0] == self.ent32.numel()
        self.space = space
        self.id_offset = int(id_offset)
        self.device = emb.device
        self._lib = nat.load_library()
        self._ws: Dict[Tuple[int, int], "torch.Tensor"] = {}
        self._ws_need: Dict[Tuple[int, int, int], Tuple[int, int]] = {}     # (batch, cut, thread) -> (tuning epoch, bytes)
        self._q_pinned = None
        self._q_dev = None
        self.shadow = None            # bf16 copy of an fp32 matrix (enable_bf16_shadow): pre-selection over half the bytes
        self.shadow_min_batch = 2     # smallest batch that goes through the shadow (enable_bf16_shadow(single_query=True): 1)
        self._io: Dict[Tuple[int, int], tuple] = {}      # (batch, k) -> device + pinned result buffers of search()
        self._last_call = None        # (batch, k, cut, through the shadow, workspace) of the last search_device
        # The blocking search() stages queries and results through per-instance buffers (pinned query, device
        # query, cached result buffers, workspaces): one caller at a time.  The reference's ExactIndex.search is
        # read-only and therefore safe under concurrent callers; this lock keeps that property for a threaded server.
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_host(cls, rows: np.ndarray, dewi: np.ndarray, ht_mean: np.ndarray, hi_mean: np.ndarray,
                  space: str = "cosine", normalize: Optional[bool] = None, device: Optional[str] = None,
                  id_offset: int = 0, chunk_rows: int = 262144) -> "DeviceCorpus":
        """Upload raw fp32 rows and payload columns; normalise on the device (A1/A2).

        Replaces N x ``ExactIndex.add`` + ``build`` (reference backends.py:394-412).  Rows
        are streamed in chunks so that the host never needs a second copy of the matrix.
        """
        torch = _torch()
        lib = nat.load_library()
        dev = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        rows = np.asarray(rows)
        if rows.ndim != 2 or rows.shape[0] == 0:
            raise ValueError("No embeddings to build index from")
        n, d = rows.shape
        if normalize is None:
            normalize = space == "cosine"
        emb = torch.empty((n, d), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            for s in range(0, n, chunk_rows):
                e = min(n, s + chunk_rows)
                blk = torch.from_numpy(np.ascontiguousarray(rows[s:e], dtype=np.float32))
                emb[s:e].copy_(blk, non_blocking=False)
            if normalize:
                nat.check(lib.dewi_normalize_rows_f32(nat.ptr(emb), nat.ptr(emb), n, d, nat.stream_ptr()))
            cols = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64)).to(dev) for c in (dewi, ht_mean, hi_mean)]
            if any(c.numel() != n for c in cols):
                raise ValueError("payload columns must have one value per row")
            dewi32 = torch.empty(n, dtype=torch.float32, device=dev)
            ent32 = torch.empty(n, dtype=torch.float32, device=dev)
            nat.check(lib.dewi_payload_soa_f64(nat.ptr(cols[0]), nat.ptr(cols[1]), nat.ptr(cols[2]), nat.ptr(dewi32),
                                               nat.ptr(ent32), n, nat.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        return cls(emb, dewi32, ent32, space, id_offset)

    def to_bf16(self) -> "DeviceCorpus":
        """The same shard with the (already normalised) matrix rounded to bf16 (config C3)."""
        torch = _torch()
        out = torch.empty(self.emb.shape, dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._lib.dewi_convert_f32_to_bf16(nat.ptr(self.emb), nat.ptr(out), self.emb.numel(),
                                                         nat.stream_ptr()))
        return DeviceCorpus(out, self.dewi32, self.ent32, self.space, self.id_offset)

    def enable_bf16_shadow(self, single_query: bool = False) -> "DeviceCorpus":
        """Keep a bf16 copy of this fp32 matrix next to it (+50 % memory).  Batches of 2 or more cosine queries then run the
        matrix-core passes over the copy as a PRE-SELECTION — half the bytes, and 256 queries per corpus pass instead of
        32 for larger batches — and re-score the candidates from the fp32 rows with the row kernels' arithmetic
        (``dewi_knn_rerank_f32_shadow``): results equal the one-query search bit for bit.  Every dim % 8 == 0 from 136 to 1536 columns, >= 64 K
        rows; any other shape simply takes the usual path.

        ``single_query=True`` sends one-query searches through the shadow as well (k <= 16: the bf16 row kernel with
        per-workgroup lists long enough for the error band, 0.24 ms instead of 0.43 at 1 M x 768; larger k: the pass with one
        active query, 0.26 ms; same re-scoring, same answers).  Off by default (+50 % memory for a path the plain scan already
        serves at the HBM rate); a query the pass refuses on an adversarial corpus is repaired inside the library call like
        any matrix-core batch (ABI 5)."""
        torch = _torch()
        if self.is_bf16:
            raise ValueError("the corpus is already bf16")
        if self.space != "cosine":
            raise ValueError("the bf16 shadow serves cosine corpora (stored rows of unit norm)")
        if self.shadow is None:
            # the error bound of the pre-selection (2^-8 of ||e|| ||q|| + accumulation) is proven for rows of norm <= 1.0001:
            # the stored form of a cosine corpus.  Checked once (NaN rows — zero embeddings — are fine: they rank first).
            worst = float(torch.nan_to_num(torch.linalg.vector_norm(self.emb, dim=1), nan=0.0).max()) if self.n_rows else 0.0
            if worst > 1.0001:
                raise ValueError(f"rows are not normalised (largest norm {worst:.6f}): the bf16 shadow's error bound "
                                 f"does not hold for this matrix")
            out = torch.empty(self.emb.shape, dtype=torch.bfloat16, device=self.device)
            with torch.cuda.device(self.device):
                nat.check(self._lib.dewi_convert_f32_to_bf16(nat.ptr(self.emb), nat.ptr(out), self.emb.numel(), nat.stream_ptr()))
            self.shadow = out
        self.shadow_min_batch = 1 if single_query else 2
        return self

    # ------------------------------------------------------------------ properties
    @property
    def n_rows(self) -> int:
        return int(self.emb.shape[0])

    @property
    def dim(self) -> int:
        return int(self.emb.shape[1])

    @property
    def is_bf16(self) -> bool:
        return self.emb.dtype == _torch().bfloat16

    def corpus_bytes(self) -> int:
        return self.emb.numel() * self.emb.element_size()

    # ------------------------------------------------------------------ helpers
    def _workspace(self, n_queries: int, n_candidates: int):
        # the required size depends on the launch plan (tunable): asked again whenever `tuning()` has been called since
        # (the library call plans every path of the shape — a few microseconds that a 40 us search need not pay each time)
        key = (n_queries, n_candidates)
        nkey = (n_queries, n_candidates, threading.get_ident())      # (the library's tuning is thread-local)
        cached = self._ws_need.get(nkey)
        if cached is not None and cached[0] == _tuning_epoch:
            need = cached[1]
        else:
            need = int(self._lib.dewi_knn_workspace_bytes(self.n_rows, self.dim, n_queries, n_candidates))
            if need == 0:
                raise nat.NativeLibraryError("dewi_knn_workspace_bytes returned 0: " + nat.last_error())
            if len(self._ws_need) > 64:
                self._ws_need.clear()
            self._ws_need[nkey] = (_tuning_epoch, need)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            if len(self._ws) > 8:
                self._ws.clear()
            ws = _torch().empty(need, dtype=_torch().uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def stage_queries(self, queries: ArrayLike):
        """Host or device query batch -> contiguous fp32 [B, d] tensor on this device."""
        torch = _torch()
        if isinstance(queries, torch.Tensor):
            q = queries
            if q.dim() == 1:
                q = q.unsqueeze(0)
            return q.to(device=self.device, dtype=torch.float32).contiguous()
        q = np.asarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        q = np.ascontiguousarray(q)
        if self._q_pinned is None or self._q_pinned.shape != q.shape:
            self._q_pinned = torch.empty(q.shape, dtype=torch.float32, pin_memory=True)
            self._q_pinned_np = self._q_pinned.numpy()
            self._q_dev = torch.empty(q.shape, dtype=torch.float32, device=self.device)
        self._q_pinned_np[...] = q
        self._q_dev.copy_(self._q_pinned, non_blocking=True)
        return self._q_dev

    # ------------------------------------------------------------------ hot path
    def search_device(self, q_dev, k: int, eta: float, entropy_pref: float, out_ids=None, out_scores=None,
                      candidates: Optional[int] = None, similarity: str = "ip", use_shadow: bool = True):
        """Enqueue one search on the current stream; returns device tensors, no sync.

        q_dev: fp32 [B, d] on this device (raw queries; cosine normalisation happens in-kernel).
        ``candidates``: size of the similarity cut that is re-ranked; default min(2k, N) as the
        reference's ExactIndex, ``candidates=k`` gives the rule of its HNSW / FAISS backends, whose
        blend uses ``similarity`` = "ip" (faiss inner product, the raw score), "one_minus_dist"
        (hnswlib: 1 - dist) or "inv_one_plus_dist" (faiss L2: 1/(1+dist)) — reference
        backends.py:229-231, 335-338.

        NOT thread-safe on one instance (shared workspaces).  Always answered: a query that a matrix-core pass of
        the batch refuses (adversarial corpora) is repaired inside the library call, on the same stream (ABI 5) — no
        id -1 ever reaches the outputs, so there is nothing for the caller to check after synchronising.
        """
        torch = _torch()
        b = int(q_dev.shape[0])
        if q_dev.shape[1] != self.dim:
            raise ValueError(f"Expected query shape ({self.dim},), got {tuple(q_dev.shape[1:])}")
        k = int(k)
        if k <= 0:
            return (torch.empty((b, 0), dtype=torch.int64, device=self.device),
                    torch.empty((b, 0), dtype=torch.float32, device=self.device))
        c = min(2 * k, self.n_rows) if candidates is None else min(int(candidates), self.n_rows)
        if out_ids is None:
            out_ids = torch.empty((b, k), dtype=torch.int64, device=self.device)
        if out_scores is None:
            out_scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
        if similarity not in nat.SIM_CODES:
            raise ValueError(f"unknown similarity {similarity!r}")
        if candidates is None and similarity != "ip":
            raise ValueError("similarity transforms belong to the ANN re-rank rule: pass candidates=k as well")
        ws = self._workspace(b, max(c, 1))
        through_shadow = candidates is None and self.shadow is not None and b >= self.shadow_min_batch and use_shadow
        self._last_call = (b, k, max(c, 1), bool(through_shadow), ws)
        if through_shadow:
            rc = self._lib.dewi_knn_rerank_f32_shadow(
                nat.ptr(self.emb), nat.ptr(self.shadow), self.n_rows, self.dim, nat.ptr(q_dev), b, nat.ptr(self.dewi32),
                nat.ptr(self.ent32), k, float(eta), float(entropy_pref), nat.SPACE_CODES[self.space], nat.ptr(out_ids),
                nat.ptr(out_scores), nat.ptr(ws), ws.numel(), nat.stream_ptr())
        elif candidates is None:
            fn = self._lib.dewi_knn_rerank_bf16 if self.is_bf16 else self._lib.dewi_knn_rerank_f32
            rc = fn(nat.ptr(self.emb), self.n_rows, self.dim, nat.ptr(q_dev), b, nat.ptr(self.dewi32),
                    nat.ptr(self.ent32), k, float(eta), float(entropy_pref), nat.SPACE_CODES[self.space],
                    nat.ptr(out_ids), nat.ptr(out_scores), nat.ptr(ws), ws.numel(), nat.stream_ptr())
        else:
            rc = self._lib.dewi_knn_rerank_candidates(
                nat.ptr(self.emb), 1 if self.is_bf16 else 0, self.n_rows, self.dim, nat.ptr(q_dev), b, nat.ptr(self.dewi32),
                nat.ptr(self.ent32), k, int(candidates), float(eta), float(entropy_pref), nat.SPACE_CODES[self.space],
                nat.SIM_CODES[similarity], nat.ptr(out_ids), nat.ptr(out_scores), nat.ptr(ws), ws.numel(), nat.stream_ptr())
        nat.check(rc)
        return out_ids, out_scores

    def refused_by_last_call(self) -> np.ndarray:
        """Monitoring / tests: bool [B] — which queries of the LAST ``search_device`` call a matrix-core pass refused (and the
        repair launches inside the same library call answered).  All False for a shape that takes the row kernels.
        Synchronises the current stream."""
        torch = _torch()
        b, k, c, through_shadow, ws = self._last_call
        off = ctypes.c_size_t(0)
        with torch.cuda.device(self.device):
            nat.check(self._lib.dewi_knn_refusal_flags(1 if self.is_bf16 else 0, 1 if through_shadow else 0, self.n_rows, self.dim,
                                                       b, k, c, nat.SPACE_CODES[self.space], ctypes.byref(off)))
            torch.cuda.current_stream().synchronize()
        if off.value == ctypes.c_size_t(-1).value:
            return np.zeros(b, dtype=bool)
        return ws[off.value: off.value + 4 * b].view(torch.int32).cpu().numpy() != 0

    def scan_kernel_name(self, n_queries: int, k: int, candidates: Optional[int] = None) -> str:
        """The kernel that streams the corpus for a batch of this size (``dewi_knn_scan_kernel``): measurement label."""
        c = min(2 * int(k), self.n_rows) if candidates is None else min(int(candidates), self.n_rows)
        buf = ctypes.create_string_buffer(128)
        with _torch().cuda.device(self.device):
            nat.check(self._lib.dewi_knn_scan_kernel(1 if self.is_bf16 else 0, self.n_rows, self.dim, int(n_queries), max(c, 1),
                                                     nat.SPACE_CODES[self.space], buf, 128))
        return buf.value.decode()

    def search(self, queries: ArrayLike, k: int = 10, eta: float = 0.5, entropy_pref: float = 0.0,
               candidates: Optional[int] = None, similarity: str = "ip") -> Tuple[np.ndarray, np.ndarray]:
        """Blocking convenience: (ids int64 [B,k] including id_offset, scores fp32 [B,k]) on the host.
        Safe to call from several threads on one instance (serialised by a per-corpus lock)."""
        torch = _torch()
        with self._lock, torch.cuda.device(self.device):
            q = self.stage_queries(queries)
            # Small result sets (one query, a handful: what the reference's search returns) are written by the select kernel
            # STRAIGHT INTO pinned host memory (device-visible at its own address on ROCm; 12 bytes per result over PCIe): no
            # device buffer, no copy command, ONE stream synchronisation — 3.5 us less per blocking call than a device
            # buffer + async D2H copy (round 4: C1 45.4 -> 41.9 us p50, C2 464.0 -> 460.5).  Large batches keep the device
            # buffer and one DMA copy (tens of thousands of 4- and 8-byte stores over PCIe would cost more than they save).
            b, kk = int(q.shape[0]), max(int(k), 0)
            io = self._io.get((b, kk))
            if io is None:
                if len(self._io) > 8:
                    self._io.clear()
                # ids (int64) and scores (fp32) share one allocation, so that they return in one copy
                n_el = b * kk
                hbuf = torch.empty(n_el * 12, dtype=torch.uint8, pin_memory=True)
                dbuf = None if n_el <= 4096 else torch.empty(n_el * 12, dtype=torch.uint8, device=self.device)
                tgt = hbuf if dbuf is None else dbuf
                io = (tgt[: n_el * 8].view(torch.int64).view(b, kk), tgt[n_el * 8:].view(torch.float32).view(b, kk),
                      hbuf[: n_el * 8].view(torch.int64).view(b, kk), hbuf[n_el * 8:].view(torch.float32).view(b, kk),
                      dbuf, hbuf)
                self._io[(b, kk)] = io
            if kk > 0:
                self.search_device(q, k, eta, entropy_pref, io[0], io[1], candidates=candidates, similarity=similarity)
                if io[4] is not None:
                    io[5].copy_(io[4], non_blocking=True)
                torch.cuda.current_stream().synchronize()
            ids_h = io[2].numpy().copy()
            scores_h = io[3].numpy().copy()
            # (a query a matrix-core pass refused was repaired inside the library call, on the stream: ABI 5)
        if self.id_offset:
            ids_h = ids_h + self.id_offset
        return ids_h, scores_h

    def candidates_device(self, q_dev, n_candidates: int, out=None):
        """Per-shard top-``n_candidates`` records, int32 view [B, n_candidates, 4] (16 B each)."""
        torch = _torch()
        b = int(q_dev.shape[0])
        if out is None:
            out = torch.empty((b, n_candidates, 4), dtype=torch.int32, device=self.device)
        c_local = max(1, min(n_candidates, self.n_rows))
        ws = self._workspace(b, c_local)
        self._last_call = (b, max(1, c_local // 2), c_local, False, ws)
        rc = self._lib.dewi_knn_candidates(nat.ptr(self.emb), 1 if self.is_bf16 else 0, self.n_rows, self.dim,
                                           nat.ptr(q_dev), b, nat.ptr(self.dewi32), nat.ptr(self.ent32),
                                           int(n_candidates), nat.SPACE_CODES[self.space], self.id_offset, nat.ptr(out),
                                           nat.ptr(ws), ws.numel(), nat.stream_ptr())
        nat.check(rc)
        return out


class PipelinedSearcher:
    """Two query batches in flight on one GPU (throughput mode).

    Batch i's scan runs on ``scan_stream``; its finish (select, blend, top-k — or, for a doc-id
    shard, the candidate records) runs on ``finish_stream`` from one of two workspaces, so it
    overlaps the scan of batch i+1.  Scans themselves stay back to back on one stream: nothing
    competes with the corpus stream for HBM.  ``submit`` only enqueues; call ``drain`` before
    reading the outputs.

    ``dewi_knn_scan`` / ``dewi_knn_finish`` take the same kernels as the one-call search (a batch of
    queries: the matrix-core passes); a query such a pass refuses is repaired by ``dewi_knn_finish`` itself
    (ABI 5), from the query tensor that was SUBMITTED — which must therefore stay untouched until the finish
    step has run (``drain``, or an event on ``finish_stream``): a caller that reuses one staging buffer for
    its queries submits a clone.  The workspace size and the path are fixed from the
    submitting thread's tuning (``_engine.tuning`` is thread-local): construct and submit from one thread.
    """

    def __init__(self, corpus: DeviceCorpus, k: int, eta: float, entropy_pref: float, n_queries: int = 1,
                 n_candidates: Optional[int] = None, finish_stream=None, depth: int = 2, scan_streams: int = 1):
        torch = _torch()
        self.corpus = corpus
        self.k, self.eta, self.pref, self.b = int(k), float(eta), float(entropy_pref), int(n_queries)
        self.c = int(n_candidates) if n_candidates is not None else min(2 * self.k, corpus.n_rows)
        self._lib = corpus._lib
        need = int(self._lib.dewi_knn_workspace_bytes(corpus.n_rows, corpus.dim, self.b, max(1, min(self.c, corpus.n_rows))))
        if need == 0:
            raise nat.NativeLibraryError("dewi_knn_workspace_bytes returned 0: " + nat.last_error())
        self._need = need
        with torch.cuda.device(corpus.device):
            # scan_streams > 1: consecutive scans alternate between streams, so the tail of one (block merge,
            # last workgroups) overlaps the ramp of the next (query preparation) — small shards only
            # (alternating priorities: streams of different priority never share a hardware queue, so the
            # overlap does not depend on how the runtime happens to map streams to queues)
            self._scan_streams = [torch.cuda.Stream(priority=-(j % 2)) for j in range(max(1, int(scan_streams)))]
            self.scan_stream = self._scan_streams[0]
            self.finish_stream = finish_stream if finish_stream is not None else torch.cuda.Stream()
            self.depth = max(2, int(depth))   # workspaces in rotation = scans that may run ahead of their finish
            self._ws = [torch.empty(need, dtype=torch.uint8, device=corpus.device) for _ in range(self.depth)]
            self._scan_done = [torch.cuda.Event() for _ in range(self.depth)]
            self._finish_done = [torch.cuda.Event() for _ in range(self.depth)]
        self._i = 0
        self._elem = 1 if corpus.is_bf16 else 0
        self._space = nat.SPACE_CODES[corpus.space]
        self._emb, self._dewi, self._ent = nat.ptr(corpus.emb), nat.ptr(corpus.dewi32), nat.ptr(corpus.ent32)
        self._s_scans = [int(st.cuda_stream) for st in self._scan_streams]
        self._s_fin = int(self.finish_stream.cuda_stream)

    def submit(self, q_dev, out_ids=None, out_scores=None, out_records=None) -> None:
        """Enqueue one query batch.  Final results go to (out_ids, out_scores); with ``out_records``
        (int32 [B, c, 4]) the shard's candidate records are written instead."""
        i = self._i
        slot = i % self.depth
        self._i = i + 1
        c = self.corpus
        which = i % len(self._scan_streams)
        scan_stream = self._scan_streams[which]
        if i >= self.depth:
            scan_stream.wait_event(self._finish_done[slot])            # workspace `slot` is free again
        rc = self._lib.dewi_knn_scan(self._emb, self._elem, c.n_rows, c.dim, q_dev.data_ptr(), self.b, self.c,
                                     self._space, self._ws[slot].data_ptr(), self._need, self._s_scans[which])
        if rc:
            nat.check(rc)
        self._scan_done[slot].record(scan_stream)
        self.finish_stream.wait_event(self._scan_done[slot])
        if out_records is None:
            rc = self._lib.dewi_knn_finish(self._ws[slot].data_ptr(), self._need, self._emb, self._elem, c.n_rows, c.dim,
                                           q_dev.data_ptr(), self.b, self.c, self._space, self.k, self.eta, self.pref, self._dewi,
                                           self._ent, c.id_offset, out_ids.data_ptr(), out_scores.data_ptr(), 0, self._s_fin)
        else:
            rc = self._lib.dewi_knn_finish(self._ws[slot].data_ptr(), self._need, self._emb, self._elem, c.n_rows, c.dim,
                                           q_dev.data_ptr(), self.b, self.c, self._space, 0, 0.0, 0.0, self._dewi, self._ent,
                                           c.id_offset, 0, 0, out_records.data_ptr(), self._s_fin)
        if rc:
            nat.check(rc)
        self._finish_done[slot].record(self.finish_stream)

    def drain(self) -> None:
        for st in self._scan_streams:
            st.synchronize()
        self.finish_stream.synchronize()


# scratch of the large merge, one buffer per (device, stream): reuse is ordered by the stream it is used on, so two streams
# (the current stream beside a PipelinedSearcher / sharded finish stream) or two threads never share one
_merge_ws: Dict[object, "torch.Tensor"] = {}


def merge_rerank_device(lists, n_candidates: int, k: int, eta: float, entropy_pref: float, out_ids=None,
                        out_scores=None):
    """lists: int32 [n_lists, B, list_len, 4] candidate records (the all-gather result)."""
    torch = _torch()
    lib = nat.load_library()
    n_lists, b, list_len = int(lists.shape[0]), int(lists.shape[1]), int(lists.shape[2])
    if out_ids is None:
        out_ids = torch.empty((b, k), dtype=torch.int64, device=lists.device)
    if out_scores is None:
        out_scores = torch.empty((b, k), dtype=torch.float32, device=lists.device)
    # up to 2048 records per query are merged in LDS; beyond that (k > 128 at eight shards) through a scratch buffer
    need = int(lib.dewi_merge_workspace_bytes(n_lists, b, list_len, int(n_candidates)))
    ws = None
    if need:
        key = (lists.device, int(torch.cuda.current_stream(lists.device).cuda_stream))
        ws = _merge_ws.get(key)
        if ws is None or ws.numel() < need:
            if len(_merge_ws) > 16:
                _merge_ws.clear()
            # (a buffer that is outgrown is only dropped here: the caching allocator keeps its memory stream-ordered, so a
            # merge still running on this stream finishes on it before anything else of this stream can take the block)
            ws = _merge_ws[key] = torch.empty(need, dtype=torch.uint8, device=lists.device)
    rc = lib.dewi_merge_rerank(nat.ptr(lists), n_lists, b, list_len, int(n_candidates), int(k), float(eta),
                               float(entropy_pref), nat.ptr(out_ids), nat.ptr(out_scores),
                               nat.ptr(ws) if ws is not None else None, need, nat.stream_ptr())
    nat.check(rc)
    return out_ids, out_scores


def records_to_numpy(recs) -> np.ndarray:
    """int32 [.., 4] record tensor -> structured host array (sim, dewi, ent, id)."""
    a = recs.detach().cpu().numpy()
    dt = np.dtype([("sim", np.float32), ("dewi", np.float32), ("ent", np.float32), ("id", np.int32)])
    return np.ascontiguousarray(a).view(dt).reshape(a.shape[:-1])


def timing(every) -> None:
    """Bracket every ``every``-th scan with hipEvents (True == 1: every scan; False / 0: off)."""
    nat.check(nat.load_library().dewi_timing_enable(int(every)))


def timing_read() -> Tuple[float, int]:
    ms = ctypes.c_double(0.0)
    n = ctypes.c_int(0)
    nat.check(nat.load_library().dewi_timing_read(ctypes.byref(ms), ctypes.byref(n)))
    return float(ms.value), int(n.value)


def prepare_queries_bf16(q_dev, space: str = "cosine"):
    """The query-preparation kernel of the batched matrix-core path on its own: normalised (cosine) bf16
    queries [B, d] on the device (``dewi_prepare_queries_bf16``).  For parity tests and diagnostics."""
    torch = _torch()
    lib = nat.load_library()
    q = q_dev.to(dtype=torch.float32).contiguous()
    out = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
    with torch.cuda.device(q.device):
        nat.check(lib.dewi_prepare_queries_bf16(nat.ptr(q), int(q.shape[0]), int(q.shape[1]), nat.SPACE_CODES[space],
                                                nat.ptr(out), nat.stream_ptr()))
    return out


_tuning_epoch = 0      # bumped by tuning(): cached workspace sizes of every DeviceCorpus are asked for again


def tuning(scan_blocks: int = 0, rows_per_iter: int = 0, nontemporal: int = -1, batched_mfma: int = 1) -> None:
    """Launch-shape overrides of the CALLING THREAD (the library keeps them thread-local).
    batched_mfma: 0 row kernels only; 1 (default) cosine batches on the matrix cores, ``space="l2"`` batches over an fp32
    corpus too (exact-refine mode: error-widened cut, candidates re-scored with the row kernels' arithmetic); 2 also l2
    batches over a bf16 corpus, unrefined (2<e,q> - ||e||^2 - ||q||^2: absolute error ~ulp(||e||^2+||q||^2) —
    near-duplicates of a query lose their near-zero distance; not the parity path)."""
    global _tuning_epoch
    _tuning_epoch += 1
    nat.check(nat.load_library().dewi_tuning_set(int(scan_blocks), int(rows_per_iter), int(nontemporal),
                                                 int(batched_mfma)))
