"""Index backends of the MI355X-native DEWI drop-in.

Same module name and public surface as reference ``src/dewi/backends.py`` —
``IndexBackend``, ``BaseIndex``, ``ExactIndex``, ``HNSWIndex``, ``FAISSIndex`` and the
``_HAS_HNSW`` / ``_HAS_FAISS`` flags — but ``ExactIndex`` is the brute-force index
re-designed for the GPU: the embedding matrix and the payload columns live in HBM and
``search`` is two HIP kernel launches (see ``csrc/knn_scan.hip`` and
``csrc/select_rerank.hip``).  There is no CPU search path: without the HIP extension or
without a GPU, ``build``/``search`` raise ``NativeLibraryError``.

The approximate backends (hnswlib / faiss graphs) are outside the hot path this package
rebuilds; their classes exist so that imports keep working and raise ``ImportError`` on
construction, which is what the reference does when those libraries are not installed
(backends.py:171-172, 249-250).
"""
from __future__ import annotations

import enum
import json
import logging
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .types import PAYLOAD_FIELDS, Payload, PayloadStore, payload_columns

logger = logging.getLogger(__name__)

# ANN libraries are not part of this build (graph indexes are out of scope).
_HAS_HNSW = False
_HAS_FAISS = False


class IndexBackend(enum.Enum):
    """Backend selector (reference backends.py:32-49).  ``AUTO`` resolves to the HIP exact index."""

    HNSW = enum.auto()
    FAISS_IVFFLAT = enum.auto()
    FAISS_HNSW = enum.auto()
    EXACT = enum.auto()

    @classmethod
    def from_str(cls, name: str) -> "IndexBackend":
        key = name.upper()
        if key == "AUTO":
            return cls.EXACT
        return cls[key]  # KeyError for unknown names, as in the reference


SearchResult = List[Tuple[str, float, Payload]]


class BaseIndex:
    """Backend contract (reference backends.py:54-163): add / build / search / save / load."""

    def __init__(self, dim: int, space: str = "cosine", **kwargs: Any):
        self.dim = dim
        self.space = space
        self._index = None
        self._doc_ids: List[str] = []
        self._payloads: Dict[str, Payload] = {}
        self._is_trained = False

    def add(self, doc_id: str, embedding: np.ndarray, payload: Payload) -> None:
        raise NotImplementedError

    def build(self, **kwargs: Any) -> None:
        raise NotImplementedError

    def search(self, query: np.ndarray, k: int = 10, eta: float = 0.5, entropy_pref: float = 0.0) -> SearchResult:
        raise NotImplementedError

    # Generic persistence: metadata + payloads only (reference backends.py:104-163, key "id").
    def save(self, path: Union[str, Path]) -> None:
        root = Path(path)
        root.mkdir(parents=True, exist_ok=True)
        with open(root / "payloads.jsonl", "w") as fh:
            for doc_id in self._doc_ids:
                fh.write(json.dumps({"id": doc_id, "payload": self._payloads[doc_id].to_dict()}) + "\n")
        with open(root / "metadata.json", "w") as fh:
            json.dump({"dim": self.dim, "space": self.space, "doc_ids": self._doc_ids,
                       "is_trained": self._is_trained, "type": type(self).__name__}, fh)

    @classmethod
    def load(cls, path: Union[str, Path], **kwargs: Any) -> "BaseIndex":
        root = Path(path)
        with open(root / "metadata.json") as fh:
            meta = json.load(fh)
        target = globals().get(meta.get("type", ""), cls)
        inst = target(dim=meta["dim"], space=meta["space"], **kwargs)
        inst._doc_ids = meta["doc_ids"]
        inst._is_trained = meta["is_trained"]
        with open(root / "payloads.jsonl") as fh:
            for line in fh:
                rec = json.loads(line)
                inst._payloads[rec["id"]] = Payload.from_dict(rec["payload"])
        return inst


class HNSWIndex(BaseIndex):
    """hnswlib graph index — not part of this build (reference backends.py:166-241)."""

    def __init__(self, dim: int, space: str = "cosine", M: int = 16, ef_construction: int = 200, **kwargs: Any):
        super().__init__(dim, space, **kwargs)
        raise ImportError("HNSW not available in the MI355X build: use ExactIndex (HIP brute force)")


class FAISSIndex(BaseIndex):
    """faiss index — not part of this build (reference backends.py:244-383)."""

    def __init__(self, dim: int, space: str = "cosine", index_type: str = "IVFFlat", nlist: int = 100, **kwargs: Any):
        super().__init__(dim, space, **kwargs)
        raise ImportError("FAISS not available in the MI355X build: use ExactIndex (HIP brute force)")


class ExactIndex(BaseIndex):
    """Exact nearest-neighbour search with DEWI re-ranking, on the GPU.

    Drop-in for reference ``ExactIndex`` (backends.py:386-556): same constructor, same
    ``add``/``build``/``search`` semantics (top-``min(2k, N)`` by similarity, then the
    eta blend, then top-k), same on-disk format.  Differences, all additive:

    * ``add_batch`` / ``search_batch`` bulk entry points (the reference is one row / one
      query per call), and ``add_batch_columns`` (payloads as arrays: no Python object per row;
      ``Payload`` objects are made on demand for the rows a search returns);
    * rows are normalised by a device kernel at ``build`` (not on the host at ``add``), so
      stored rows can differ from the reference's in the last fp32 bit;
    * the payload values the re-rank reads (``dewi``, ``ht_mean``, ``hi_mean``) are
      snapshotted into HBM columns at ``build``; call ``refresh_payloads()`` after mutating
      ``Payload`` objects in place.
    """

    def __init__(self, dim: int, space: str = "cosine", **kwargs: Any):
        super().__init__(dim, space, **kwargs)
        self._payloads = PayloadStore()            # a dict (reference contract) that also serves column-ingested rows
        self._normalize = space == "cosine"
        self._pending: List[np.ndarray] = []      # raw fp32 rows (or [m, d] blocks) not yet on the device
        self._pending_rows = 0
        self._corpus = None                        # _engine.DeviceCorpus once built
        self._host_rows: Optional[np.ndarray] = None  # lazily materialised copy of the stored matrix
        self._loaded_rows: Optional[np.ndarray] = None  # rows read by load(): already in stored form
        self._device: Optional[str] = kwargs.get("device")
        # additive: keep a bf16 shadow copy of the fp32 matrix (+50 % HBM): search_batch then runs a matrix-core pass over the
        # copy as a pre-selection (half the bytes; 256 queries per corpus pass above 32) and re-scores the candidates from the
        # fp32 rows — same ids and scores as the one-query search (DeviceCorpus.enable_bf16_shadow)
        self._batch_shadow: bool = bool(kwargs.get("batch_shadow", False))
        # with batch_shadow: search() of ONE query goes through the shadow too (same answers; see enable_bf16_shadow)
        self._shadow_single: bool = bool(kwargs.get("shadow_single_query", False))

    # ---------------------------------------------------------------- ingest (A1)
    def add(self, doc_id: str, embedding: np.ndarray, payload: Payload) -> None:
        if embedding.shape != (self.dim,):
            raise ValueError(f"Expected embedding of shape {(self.dim,)}, got {embedding.shape}")
        self._doc_ids.append(doc_id)
        self._payloads[doc_id] = payload
        self._pending.append(np.array(embedding, dtype=np.float32))
        self._pending_rows += 1
        self._invalidate()

    def add_batch(self, doc_ids: Sequence[str], embeddings: np.ndarray, payloads: Sequence[Payload]) -> None:
        """Bulk ``add``: ``embeddings`` is [m, dim]; one shape check, no per-row Python work."""
        emb = np.asarray(embeddings)
        if emb.ndim != 2 or emb.shape[1] != self.dim:
            raise ValueError(f"Expected embeddings of shape (m, {self.dim}), got {emb.shape}")
        if not (len(doc_ids) == emb.shape[0] == len(payloads)):
            raise ValueError("doc_ids, embeddings and payloads must have the same length")
        self._doc_ids.extend(doc_ids)
        self._payloads.update(zip(doc_ids, payloads))
        self._pending.append(np.array(emb, dtype=np.float32))
        self._pending_rows += emb.shape[0]
        self._invalidate()

    def add_batch_columns(self, doc_ids: Sequence[str], embeddings: np.ndarray, columns: Dict[str, np.ndarray],
                          copy: bool = False) -> None:
        """Bulk ``add`` with the payloads as structure-of-arrays: ``columns[name]`` is one
        float array per ``Payload`` field (missing fields are 0.0, as ``Payload()``'s defaults).  Nothing
        is done per row in Python: a 1 M-row ingest is a list extend plus array bookkeeping, and
        ``build`` uploads the columns as they are.  ``Payload`` objects are created lazily — for the
        rows a search returns, or on ``get_payload`` / iteration over ``_payloads`` — and the same
        object is returned from then on.  ``copy=False``: the embedding block is referenced, not
        copied, until ``build()``; do not modify it in between.

        DEVICE-RESIDENT ingest: ``embeddings`` may be a CUDA fp32 tensor [m, dim] and the columns CUDA tensors (the
        ``dewi32`` column of ``DewiScorer.score_batch_device``, signal columns ...).  Nothing is copied to the host;
        when the block is the whole corpus ``build()`` uses it IN PLACE — cosine rows are normalised inside the
        caller's tensor (``copy=True`` keeps the caller's tensor intact at the price of a device copy) — and the
        HBM payload columns are computed on the device from the given columns."""
        from . import _native as nat
        if nat.is_device_tensor(embeddings):
            return self._add_device_block(doc_ids, embeddings, columns, copy)
        emb = np.asarray(embeddings)
        if emb.ndim != 2 or emb.shape[1] != self.dim:
            raise ValueError(f"Expected embeddings of shape (m, {self.dim}), got {emb.shape}")
        if len(doc_ids) != emb.shape[0]:
            raise ValueError("doc_ids and embeddings must have the same length")
        unknown = [name for name in columns if name not in PAYLOAD_FIELDS]
        if unknown:
            raise ValueError(f"unknown payload columns {unknown}")
        row0 = len(self._doc_ids)
        self._payloads.add_columns(row0, doc_ids, columns)         # validates the column shapes
        self._doc_ids.extend(doc_ids)
        self._pending.append(np.array(emb, dtype=np.float32) if copy else np.ascontiguousarray(emb, dtype=np.float32))
        self._pending_rows += emb.shape[0]
        self._invalidate()

    def _add_device_block(self, doc_ids: Sequence[str], embeddings, columns, copy: bool) -> None:
        import torch
        emb = embeddings
        if emb.dim() != 2 or int(emb.shape[1]) != self.dim:
            raise ValueError(f"Expected embeddings of shape (m, {self.dim}), got {tuple(emb.shape)}")
        if len(doc_ids) != int(emb.shape[0]):
            raise ValueError("doc_ids and embeddings must have the same length")
        unknown = [name for name in columns if name not in PAYLOAD_FIELDS]
        if unknown:
            raise ValueError(f"unknown payload columns {unknown}")
        if emb.dtype != torch.float32 or not emb.is_contiguous():
            emb = emb.to(dtype=torch.float32).contiguous()                 # a device copy: no longer the caller's tensor
        elif copy:
            emb = emb.clone()
        row0 = len(self._doc_ids)
        self._payloads.add_columns(row0, doc_ids, columns)
        self._doc_ids.extend(doc_ids)
        self._pending.append(emb)
        self._pending_rows += int(emb.shape[0])
        self._invalidate()

    def _invalidate(self) -> None:
        self._is_trained = False
        self._host_rows = None

    # ---------------------------------------------------------------- build (A2)
    def _raw_matrix(self) -> Tuple[np.ndarray, int]:
        """(all rows as one [N, d] fp32 array, number of leading rows already in stored form)."""
        blocks: List[np.ndarray] = []
        done = 0
        if self._corpus is not None:
            blocks.append(self._stored_rows())
            done = blocks[0].shape[0]
        elif self._loaded_rows is not None:
            blocks.append(self._loaded_rows)
            done = self._loaded_rows.shape[0]
        blocks.extend((b.detach().cpu().numpy() if hasattr(b, "is_cuda") else b).reshape(-1, self.dim) for b in self._pending)
        if not blocks:
            return np.empty((0, self.dim), np.float32), 0
        return (blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=0)), done

    def _build_device(self) -> None:
        """``build`` when device-resident blocks are pending: the matrix is assembled (or, for one block that is the
        whole corpus, simply taken) on the GPU, new rows are normalised there, and the two fp32 payload columns the
        re-rank reads are computed on the device.  No row and no column visits the host."""
        import torch
        from . import _native as nat
        from ._engine import DeviceCorpus
        lib = nat.load_library()
        n = len(self._doc_ids)
        dev_blocks = [b for b in self._pending if hasattr(b, "is_cuda")]
        dev = dev_blocks[0].device
        stored = None
        if self._corpus is not None:
            stored = self._corpus.emb.float() if self._corpus.is_bf16 else self._corpus.emb
        elif self._loaded_rows is not None:
            stored = torch.from_numpy(self._loaded_rows).to(dev)
        already = 0 if stored is None else int(stored.shape[0])
        with torch.cuda.device(dev):
            if stored is None and len(self._pending) == 1 and int(dev_blocks[0].shape[0]) == n:
                emb = dev_blocks[0]                                        # the caller's tensor, in place
            else:
                emb = torch.empty((n, self.dim), dtype=torch.float32, device=dev)
                at = 0
                if stored is not None:
                    emb[:already].copy_(stored)
                    at = already
                for b in self._pending:
                    blk = b if hasattr(b, "is_cuda") else torch.from_numpy(np.ascontiguousarray(b.reshape(-1, self.dim), dtype=np.float32))
                    emb[at:at + int(blk.shape[0])].copy_(blk)
                    at += int(blk.shape[0])
                if at != n:
                    raise ValueError(f"{n} doc ids but {at} embedding rows")
            if self._normalize and already < n:
                tail = emb[already:]
                nat.check(lib.dewi_normalize_rows_f32(nat.ptr(tail), nat.ptr(tail), n - already, self.dim, nat.stream_ptr()))
            # payload columns: device blocks contribute on the device; rows that came with host data through the host
            fields = ("dewi", "ht_mean", "hi_mean")
            blocks = list(self._payloads.column_blocks())
            covered = sum(r1 - r0 for r0, r1, cols, made in blocks
                          if any(hasattr(c, "is_cuda") for c in cols.values()) and not made)
            if covered == n:
                c64 = {}
                for name in fields:
                    # (a block may mix CUDA and host columns: PayloadStore keeps host ones as ndarrays — as_tensor takes both)
                    parts = [(torch.as_tensor(cols[name]).to(device=dev, dtype=torch.float64) if name in cols
                              else torch.zeros(r1 - r0, dtype=torch.float64, device=dev)) for r0, r1, cols, _ in blocks]
                    c64[name] = (parts[0] if len(parts) == 1 else torch.cat(parts)).contiguous()
            else:
                host = self._payload_columns()
                c64 = {name: torch.from_numpy(host[name]).to(dev) for name in fields}
            dewi32 = torch.empty(n, dtype=torch.float32, device=dev)
            ent32 = torch.empty(n, dtype=torch.float32, device=dev)
            nat.check(lib.dewi_payload_soa_f64(nat.ptr(c64["dewi"]), nat.ptr(c64["ht_mean"]), nat.ptr(c64["hi_mean"]),
                                               nat.ptr(dewi32), nat.ptr(ent32), n, nat.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        self._corpus = DeviceCorpus(emb, dewi32, ent32, self.space)
        if self._batch_shadow and self.space == "cosine":
            self._corpus.enable_bf16_shadow(single_query=self._shadow_single)
        self._pending = []
        self._pending_rows = 0
        self._loaded_rows = None
        self._host_rows = None
        self._is_trained = True

    def build(self, **kwargs: Any) -> None:
        from ._engine import DeviceCorpus
        if any(hasattr(b, "is_cuda") for b in self._pending):
            return self._build_device()
        rows, already = self._raw_matrix()
        if rows.shape[0] == 0:
            raise ValueError("No embeddings to build index from")
        if rows.shape[0] != len(self._doc_ids):
            raise ValueError(f"{len(self._doc_ids)} doc ids but {rows.shape[0]} embedding rows")
        cols = self._payload_columns()
        if already == 0 or not self._normalize:
            corpus = DeviceCorpus.from_host(rows, cols["dewi"], cols["ht_mean"], cols["hi_mean"], self.space,
                                            normalize=self._normalize, device=self._device)
        else:
            # rows [0, already) are stored (normalised) rows: normalise only the new tail
            head = DeviceCorpus.from_host(rows, cols["dewi"], cols["ht_mean"], cols["hi_mean"], self.space,
                                          normalize=False, device=self._device)
            if already < rows.shape[0]:
                import torch
                from . import _native as nat
                tail = head.emb[already:]
                with torch.cuda.device(head.device):       # the kernel goes on THAT device's current stream
                    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(tail), nat.ptr(tail),
                                                                         rows.shape[0] - already, self.dim,
                                                                         nat.stream_ptr()))
                    torch.cuda.current_stream().synchronize()
            corpus = head
        self._corpus = (corpus.enable_bf16_shadow(single_query=self._shadow_single)
                        if self._batch_shadow and self.space == "cosine" else corpus)
        self._pending = []
        self._pending_rows = 0
        self._loaded_rows = None
        self._host_rows = None
        self._is_trained = True

    def _payload_columns(self) -> Dict[str, np.ndarray]:
        """float64 ``dewi`` / ``ht_mean`` / ``hi_mean`` of every row, in row order.  Column-ingested
        blocks are copied as arrays (objects already handed out for some of their rows win, so in-place
        edits of a returned ``Payload`` are seen by ``refresh_payloads``); the rest is read from objects."""
        fields = ("dewi", "ht_mean", "hi_mean")
        n = len(self._doc_ids)
        store = self._payloads
        out = {name: np.zeros(n, dtype=np.float64) for name in fields}
        covered = np.zeros(n, dtype=bool)
        for row0, row1, cols, made in store.column_blocks():
            for name in fields:
                if name in cols:
                    c = cols[name]
                    out[name][row0:row1] = c.detach().cpu().numpy() if hasattr(c, "is_cuda") else c
            for row, p in made.items():
                for name in fields:
                    out[name][row] = getattr(p, name)
            covered[row0:row1] = True
        if not covered.all():
            rows = np.nonzero(~covered)[0]
            plist = [dict.__getitem__(store, self._doc_ids[r]) for r in rows.tolist()]
            sub = payload_columns(plist, fields)
            for name in fields:
                out[name][rows] = sub[name]
        return out

    def refresh_payloads(self) -> None:
        """Re-snapshot ``dewi`` / ``ht_mean`` / ``hi_mean`` of every Payload into the HBM columns."""
        if self._corpus is None:
            return
        import torch
        from . import _native as nat
        cols = self._payload_columns()
        dev = self._corpus.device
        with torch.cuda.device(dev):                       # the kernel goes on THAT device's current stream
            d = [torch.from_numpy(cols[k]).to(dev) for k in ("dewi", "ht_mean", "hi_mean")]
            nat.check(nat.load_library().dewi_payload_soa_f64(nat.ptr(d[0]), nat.ptr(d[1]), nat.ptr(d[2]),
                                                              nat.ptr(self._corpus.dewi32), nat.ptr(self._corpus.ent32),
                                                              len(self._doc_ids), nat.stream_ptr()))
            torch.cuda.current_stream().synchronize()

    # ---------------------------------------------------------------- stored rows on the host
    def _stored_rows(self) -> np.ndarray:
        if self._host_rows is None:
            self._host_rows = self._corpus.emb.float().cpu().numpy()
        return self._host_rows

    @property
    def _embeddings(self):
        """What the reference keeps in ``_embeddings``: a list of rows before ``build``, the N x d
        fp32 matrix after (index.py:101-116 reaches into it).  Materialised from HBM on demand."""
        if self._corpus is not None and not self._pending:
            return self._stored_rows()
        if self._corpus is None and self._loaded_rows is not None and not self._pending:
            return self._loaded_rows
        rows, _ = self._raw_matrix()
        return [r for r in rows]

    @_embeddings.setter
    def _embeddings(self, value) -> None:
        arr = np.asarray(value, dtype=np.float32)
        self._corpus = None
        self._pending = []
        self._pending_rows = 0
        self._loaded_rows = arr.reshape(-1, self.dim) if arr.size else None
        self._invalidate()

    # ---------------------------------------------------------------- search (A3 + A4)
    def _ensure_built(self) -> None:
        if self._corpus is None or self._pending or not self._is_trained:
            self.build()

    def search(self, query: np.ndarray, k: int = 10, eta: float = 0.5, entropy_pref: float = 0.0,
               candidates: Optional[int] = None, similarity: str = "ip") -> SearchResult:
        """Reference ``ExactIndex.search`` (backends.py:414-481) for one query.

        ``candidates`` (additive): how many nearest rows are re-ranked.  Default ``min(2k, N)``, the
        reference's ExactIndex rule; ``candidates=k`` is the rule of its HNSW / FAISS backends
        (backends.py:217-240, 326-356: exactly k neighbours are blended and sorted) on exact neighbours,
        with ``similarity`` choosing what those backends blend: "ip" (faiss inner product, :335-336),
        "one_minus_dist" (hnswlib ``1 - dist``, :229-231) or "inv_one_plus_dist" (faiss L2
        ``1/(1+dist)``, :337-338).  hnswlib / faiss are not installed here: that rule is restated by
        reading and its parity is unpinned.
        """
        q = np.asarray(query, dtype=np.float32)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        rows, scores = self.search_batch(q, k, eta, entropy_pref, candidates, similarity)
        return self.results_for(rows[:1], scores[:1])[0]

    def search_batch(self, queries: np.ndarray, k: int = 10, eta: float = 0.5, entropy_pref: float = 0.0,
                     candidates: Optional[int] = None, similarity: str = "ip") -> Tuple[np.ndarray, np.ndarray]:
        """[B, dim] queries -> (row indices int64 [B, k], adjusted scores fp32 [B, k])."""
        self._ensure_built()
        q = np.asarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"Expected queries of shape (B, {self.dim}), got {q.shape}")
        return self._corpus.search(q, int(k), float(eta), float(entropy_pref), candidates=candidates,
                                   similarity=similarity)

    def results_for(self, rows: np.ndarray, scores: np.ndarray) -> List[SearchResult]:
        """Row indices/scores of ``search_batch`` -> the reference's (doc_id, score, Payload) tuples."""
        ids, at_row = self._doc_ids, self._payloads.at_row
        if self._payloads._blocks:      # column blocks (possibly device-resident): one gather for the rows that are wanted
            self._payloads.ensure_rows(np.unique(np.asarray(rows)).tolist(), ids)
        return [[(ids[r], float(s), at_row(r, ids[r])) for r, s in zip(rr.tolist(), ss.tolist())]
                for rr, ss in zip(rows, scores)]

    # ---------------------------------------------------------------- persistence (reference :483-556)
    def save(self, path: Union[str, Path]) -> None:
        root = Path(path)
        root.mkdir(parents=True, exist_ok=True)
        if self._pending:
            # the file format stores rows in normalised form (the reference normalises at add());
            # here normalisation is a device kernel, so pending rows are built first
            self.build()
        emb = self._embeddings
        matrix = emb if isinstance(emb, np.ndarray) else (np.array(emb) if len(emb) else np.empty((0, self.dim), np.float32))
        with open(root / "metadata.json", "w") as fh:
            json.dump({"dim": self.dim, "space": self.space, "doc_ids": self._doc_ids, "normalize": self._normalize,
                       "is_trained": self._is_trained, "num_embeddings": int(len(matrix))}, fh)
        with open(root / "payloads.jsonl", "w") as fh:
            for doc_id in self._doc_ids:
                fh.write(json.dumps({"doc_id": doc_id, "payload": self._payloads[doc_id].to_dict()}) + "\n")
        if len(matrix) > 0:
            np.save(str(root / "embeddings.npy"), matrix)

    @classmethod
    def load(cls, path: Union[str, Path], **kwargs: Any) -> "ExactIndex":
        root = Path(path)
        with open(root / "metadata.json", "r") as fh:
            meta = json.load(fh)
        inst = cls(dim=meta["dim"], space=meta["space"], **kwargs)
        inst._doc_ids = meta["doc_ids"]
        inst._normalize = meta["normalize"]
        with open(root / "payloads.jsonl", "r") as fh:
            for line in fh:
                rec = json.loads(line)
                inst._payloads[rec["doc_id"]] = Payload.from_dict(rec["payload"])
        emb_path = root / "embeddings.npy"
        if emb_path.exists() and meta.get("num_embeddings", 0) > 0:
            # rows on disk are already in stored (normalised) form — whether the reference saved them
            # before or after build() — so they are uploaded as they are
            inst._loaded_rows = np.ascontiguousarray(np.load(str(emb_path), allow_pickle=False), dtype=np.float32)
        elif inst._doc_ids:
            logger.warning("No embeddings found during load, index will need to be rebuilt")
        inst._is_trained = False  # the device copy is rebuilt lazily on the first search
        return inst
