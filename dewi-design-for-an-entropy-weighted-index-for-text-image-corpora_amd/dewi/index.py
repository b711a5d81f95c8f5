"""``DewiIndex`` — the façade users construct (reference ``src/dewi/index.py``).

Same constructor, defaults and methods as the reference (index.py:22-166): it validates
the query, fills in the default ``eta`` / ``entropy_pref`` and delegates to a backend.  In
this build every backend choice resolves to the HIP ``ExactIndex`` (ANN graph libraries
are out of scope), with the reference's own warning when an ANN backend was asked for.
Additions: ``add_batch`` and ``search_batch``.
"""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import backends as _backends
from .backends import (BaseIndex, ExactIndex, FAISSIndex, HNSWIndex, IndexBackend, _HAS_FAISS, _HAS_HNSW)
from .types import Payload

logger = logging.getLogger(__name__)


class DewiIndex(BaseIndex):
    def __init__(self, dim: int, space: str = "cosine", backend: Union[str, IndexBackend] = "auto", ef: int = 200,
                 M: int = 32, use_ann: bool = True, ef_query: int = 200, rerank_eta: float = 0.25,
                 entropy_pref: float = 0.0, **kwargs: Any):
        super().__init__(dim, space)
        self._meta: Dict[str, Dict[str, Any]] = {}
        self.ef_query = ef_query
        self.rerank_eta = float(rerank_eta)
        self.entropy_pref = float(entropy_pref)
        self._built = False
        self._use_ann = bool(use_ann)
        if isinstance(backend, str):
            try:
                backend = IndexBackend.from_str(backend)
            except KeyError:
                backend = IndexBackend.EXACT          # unknown names fall back (index.py:44-48)
        self._requested_backend = backend
        if self._use_ann and backend is not IndexBackend.EXACT:
            # index.py:58-60: requested ANN library missing -> warn, use the exact index
            logger.warning("ANN backend unavailable; falling back to ExactIndex.")
        # (additive switches of the exact backend; everything else in **kwargs is accepted and ignored as in the reference)
        exact_kwargs = {k: v for k, v in kwargs.items() if k in ("device", "batch_shadow", "shadow_single_query")}
        self._backend: BaseIndex = ExactIndex(dim, space, **exact_kwargs)

    # ------------------------------------------------------------------ ingest / build
    def add(self, doc_id: str, embedding: np.ndarray, payload: Payload, meta: Optional[Dict[str, Any]] = None) -> None:
        if meta is not None:
            self._meta[doc_id] = meta
        self._backend.add(doc_id, np.asarray(embedding, dtype=np.float32), payload)
        self._built = False

    def add_batch(self, doc_ids: Sequence[str], embeddings: np.ndarray, payloads: Sequence[Payload]) -> None:
        self._backend.add_batch(doc_ids, np.asarray(embeddings, dtype=np.float32), payloads)
        self._built = False

    def add_batch_columns(self, doc_ids: Sequence[str], embeddings: np.ndarray, columns: Dict[str, np.ndarray]) -> None:
        """Bulk ingest with the payloads as one float array per ``Payload`` field (no object per row)."""
        self._backend.add_batch_columns(doc_ids, embeddings, columns)
        self._built = False

    def build(self) -> None:
        self._backend.build()
        self._built = True

    # ------------------------------------------------------------------ search (A5)
    def _defaults(self, eta: Optional[float], entropy_pref: Optional[float]) -> Tuple[float, float]:
        return (self.rerank_eta if eta is None else eta, self.entropy_pref if entropy_pref is None else entropy_pref)

    def search(self, query: np.ndarray, k: int = 10, eta: Optional[float] = None,
               entropy_pref: Optional[float] = None) -> List[Tuple[str, float, Payload]]:
        if not self._built:
            self.build()
        eta, entropy_pref = self._defaults(eta, entropy_pref)
        q = np.asarray(query, dtype=np.float32)
        if q.shape != (self.dim,):
            raise ValueError(f"Expected query shape ({self.dim},), got {q.shape}")
        return self._backend.search(q, k, eta, entropy_pref)

    def search_batch(self, queries: np.ndarray, k: int = 10, eta: Optional[float] = None,
                     entropy_pref: Optional[float] = None) -> List[List[Tuple[str, float, Payload]]]:
        """One call for B queries ([B, dim]); each result list equals ``search`` of that row."""
        if not self._built:
            self.build()
        eta, entropy_pref = self._defaults(eta, entropy_pref)
        q = np.asarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"Expected queries of shape (B, {self.dim}), got {q.shape}")
        rows, scores = self._backend.search_batch(q, k, eta, entropy_pref)
        return self._backend.results_for(rows, scores)

    # ------------------------------------------------------------------ accessors (index.py:95-119)
    def __len__(self) -> int:
        return len(self._backend._doc_ids)

    def get_payload(self, doc_id: str) -> Optional[Payload]:
        return self._backend._payloads.get(doc_id)

    def get_embedding(self, doc_id: str) -> Optional[np.ndarray]:
        store = getattr(self._backend, "_embeddings", None)
        if store is None:
            return None
        try:
            return store[self._backend._doc_ids.index(doc_id)]
        except (ValueError, IndexError):
            return None

    def get_metadata(self, doc_id: str) -> Optional[Dict[str, Any]]:
        return self._meta.get(doc_id)

    # ------------------------------------------------------------------ persistence (index.py:121-166)
    def save(self, path: Union[str, Path]) -> None:
        root = Path(path)
        root.mkdir(parents=True, exist_ok=True)
        self._backend.save(root / "ann_index")
        cfg = {"dim": self.dim, "space": self.space, "use_ann": self._use_ann, "ef_query": self.ef_query,
               "rerank_eta": self.rerank_eta, "entropy_pref": self.entropy_pref, "built": self._built,
               "backend_type": type(self._backend).__name__}
        with open(root / "config.json", "w", encoding="utf-8") as fh:
            json.dump(cfg, fh)
        if self._meta:
            with open(root / "meta.json", "w", encoding="utf-8") as fh:
                json.dump(self._meta, fh)

    @classmethod
    def load(cls, path: Union[str, Path]) -> "DewiIndex":
        root = Path(path)
        with open(root / "config.json", "r", encoding="utf-8") as fh:
            cfg = json.load(fh)
        backend_cls = getattr(_backends, cfg.get("backend_type", "ExactIndex"), ExactIndex)
        if backend_cls in (HNSWIndex, FAISSIndex):
            backend_cls = ExactIndex  # graph files are not readable here; the exact index serves the same rows
        inst = cls(dim=cfg["dim"], space=cfg["space"], backend="exact", use_ann=cfg.get("use_ann", True),
                   ef_query=cfg.get("ef_query", 200), rerank_eta=cfg.get("rerank_eta", 0.25),
                   entropy_pref=cfg.get("entropy_pref", 0.0))
        inst._backend = backend_cls.load(root / "ann_index")
        inst._built = False  # device copy is rebuilt on first use
        meta_path = root / "meta.json"
        if meta_path.exists():
            with open(meta_path, "r", encoding="utf-8") as fh:
                inst._meta = json.load(fh)
        return inst


__all__ = ["DewiIndex", "BaseIndex", "ExactIndex", "HNSWIndex", "FAISSIndex", "IndexBackend", "_HAS_FAISS",
           "_HAS_HNSW", "Payload"]
