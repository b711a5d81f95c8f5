"""Data contract of the DEWI hot path: the per-document ``Payload`` record and the
scorer ``Weights``.

Mirrors reference ``src/dewi/types.py:8-51`` field for field (names, order, defaults,
JSON codecs) so that objects, ``payloads.jsonl`` files and keyword construction are
interchangeable with the reference.  Additions are the structure-of-arrays helpers the
device path needs: the index keeps payloads as fp32/f64 columns in HBM, not as objects.
"""
from __future__ import annotations

import dataclasses
import json
import bisect
from typing import Dict, Iterable, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np

#: The seven signals the scorer standardises, in the order the C ABI expects
#: (include/dewi_hip.h, DEWI_NUM_SIGNALS).
SIGNAL_FIELDS = ("ht_mean", "ht_q90", "hi_mean", "hi_q90", "I_hat", "redundancy", "noise")


@dataclasses.dataclass
class Payload:
    """Scores and signals attached to one indexed document (reference types.py:8-19)."""

    dewi: float = 0.0
    ht_mean: float = 0.0
    ht_q90: float = 0.0
    hi_mean: float = 0.0
    hi_q90: float = 0.0
    I_hat: float = 0.0
    redundancy: float = 0.0
    noise: float = 0.0

    # -- codecs (reference types.py:21-39) ------------------------------------------------
    def to_dict(self) -> Dict[str, float]:
        return {name: getattr(self, name) for name in PAYLOAD_FIELDS}

    @classmethod
    def from_dict(cls, data: Mapping[str, float]) -> "Payload":
        """Unknown keys are dropped, values go through ``float()``."""
        return cls(**{name: float(data[name]) for name in PAYLOAD_FIELDS if name in data})

    def to_bytes(self) -> bytes:
        return json.dumps(self.to_dict()).encode("utf-8")

    @classmethod
    def from_bytes(cls, data: bytes) -> "Payload":
        return cls.from_dict(json.loads(data.decode("utf-8")))


PAYLOAD_FIELDS = tuple(f.name for f in dataclasses.fields(Payload))


@dataclasses.dataclass
class Weights:
    """Scorer weights (reference types.py:42-51)."""

    alpha_t: float = 1.0
    alpha_i: float = 1.0
    alpha_m: float = 1.0
    alpha_r: float = 1.0
    alpha_n: float = 1.0
    delta: float = 3.0

    def as_vector(self) -> np.ndarray:
        """(alpha_t, alpha_i, alpha_m, alpha_r, alpha_n) as float64, the C-ABI order."""
        return np.array([self.alpha_t, self.alpha_i, self.alpha_m, self.alpha_r, self.alpha_n], dtype=np.float64)


def payload_columns(payloads: Sequence[Payload], fields: Iterable[str] = PAYLOAD_FIELDS) -> Dict[str, np.ndarray]:
    """Array-of-structs -> float64 columns (one pass per field)."""
    return {name: np.fromiter((getattr(p, name) for p in payloads), dtype=np.float64, count=len(payloads))
            for name in fields}


def payloads_from_columns(columns: Mapping[str, np.ndarray]) -> list:
    """float columns -> list of ``Payload`` (missing fields keep their 0.0 default)."""
    names = [n for n in PAYLOAD_FIELDS if n in columns]
    n = len(next(iter(columns.values()))) if columns else 0
    cols = [np.asarray(columns[name], dtype=np.float64).tolist() for name in names]
    return [Payload(**{name: col[i] for name, col in zip(names, cols)}) for i in range(n)]


class PayloadStore(dict):
    """``doc_id -> Payload`` as the reference's ``BaseIndex._payloads`` (backends.py:63), for corpora
    that may have been ingested as COLUMNS (``ExactIndex.add_batch_columns``).

    Documents added with a ``Payload`` object are plain dict entries (the object the caller passed is
    the object every lookup returns).  Column-ingested documents have no object until somebody asks
    for one: the first lookup builds a ``Payload`` from the columns and keeps it, so later lookups
    return the same object.  Bulk views (iteration, ``items``, ``values``, ``len`` is cheap) materialise
    whatever is still missing.
    """

    def __init__(self) -> None:
        super().__init__()
        self._starts: List[int] = []          # first global row of each column block
        self._blocks: List[Tuple[int, int, Sequence[str], Dict[str, np.ndarray]]] = []   # (row0, row1, ids, columns)
        self._made: List[Dict[int, "Payload"]] = []      # per block: rows that already have an object
        self._id_rows: Optional[Dict[str, int]] = None    # doc_id -> global row of column-ingested docs (lazy)
        self._lazy_count = 0

    # ---- ingest -----------------------------------------------------------------------------
    def add_columns(self, row0: int, doc_ids: Sequence[str], columns: Mapping[str, np.ndarray]) -> None:
        """``columns[name]``: a host array, or a CUDA tensor that STAYS on the device (``ExactIndex`` builds its HBM
        columns from it there); ``Payload`` objects of such a block are made from the handful of rows a search
        returns (one gather + one small copy, ``ensure_rows``) — only whole-store views (iteration, ``save``) bring a
        device block to the host."""
        cols = {name: (columns[name] if getattr(columns[name], "is_cuda", False) else np.asarray(columns[name], dtype=np.float64))
                for name in PAYLOAD_FIELDS if name in columns}
        n = len(doc_ids)
        for name, c in cols.items():
            if tuple(c.shape) != (n,):
                raise ValueError(f"payload column {name!r} has shape {tuple(c.shape)}, expected ({n},)")
        self._starts.append(int(row0))
        self._blocks.append((int(row0), int(row0) + n, doc_ids, cols))
        self._made.append({})
        self._id_rows = None
        self._lazy_count += n

    # ---- row access (search results: no doc_id -> row map needed) -----------------------------
    def _block_of(self, row: int) -> int:
        j = bisect.bisect_right(self._starts, row) - 1
        if j < 0 or row >= self._blocks[j][1]:
            return -1
        return j

    def at_row(self, row: int, doc_id: str) -> "Payload":
        hit = dict.get(self, doc_id)
        if hit is not None:
            return hit
        j = self._block_of(row)
        if j < 0:
            raise KeyError(doc_id)
        return self._make(j, row, doc_id)

    def _block_on_device(self, j: int) -> bool:
        return any(getattr(c, "is_cuda", False) for c in self._blocks[j][3].values())

    def _block_to_host(self, j: int) -> None:
        row0, row1, ids, cols = self._blocks[j]
        self._blocks[j] = (row0, row1, ids, {name: (c.detach().cpu().numpy().astype(np.float64) if getattr(c, "is_cuda", False) else c)
                                             for name, c in cols.items()})

    def ensure_rows(self, rows: Sequence[int], doc_ids: Sequence[str]) -> None:
        """Make the ``Payload`` objects of ``rows`` that live in device-resident column blocks: per block ONE gather
        of the wanted rows over all its columns and one small device-to-host copy."""
        want: Dict[int, List[int]] = {}
        for row in rows:
            if dict.__contains__(self, doc_ids[row]):
                continue
            j = self._block_of(row)
            if j >= 0 and row not in self._made[j] and self._block_on_device(j):
                want.setdefault(j, []).append(row)
        if not want:
            return
        import torch
        for j, rws in want.items():
            row0, _, _, cols = self._blocks[j]
            rws = sorted(set(rws))
            names = list(cols)
            dev = next(c.device for c in cols.values() if getattr(c, "is_cuda", False))
            idx = torch.tensor([r - row0 for r in rws], dtype=torch.int64, device=dev)
            block = torch.stack([(c if getattr(c, "is_cuda", False) else torch.from_numpy(c).to(dev)).index_select(0, idx).double()
                                 for c in cols.values()]).cpu().numpy()
            for t, row in enumerate(rws):
                p = Payload(**{name: float(block[f, t]) for f, name in enumerate(names)})
                dict.__setitem__(self, doc_ids[row], p)
                self._made[j][row] = p
                self._lazy_count -= 1

    def _make(self, j: int, row: int, doc_id: str) -> "Payload":
        row0, _, _, cols = self._blocks[j]
        p = Payload(**{name: float(c[row - row0]) for name, c in cols.items()})
        dict.__setitem__(self, doc_id, p)
        self._made[j][row] = p
        self._lazy_count -= 1
        return p

    def column_blocks(self) -> Iterator[Tuple[int, int, Dict[str, np.ndarray], Dict[int, "Payload"]]]:
        """(row0, row1, columns, objects already handed out for rows of the block)."""
        for (row0, row1, _, cols), made in zip(self._blocks, self._made):
            yield row0, row1, cols, made

    # ---- mapping protocol -----------------------------------------------------------------------
    def _rows_by_id(self) -> Dict[str, int]:
        if self._id_rows is None:
            m: Dict[str, int] = {}
            for row0, _, ids, _ in self._blocks:
                m.update(zip(ids, range(row0, row0 + len(ids))))
            self._id_rows = m
        return self._id_rows

    def __missing__(self, doc_id: str) -> "Payload":
        row = self._rows_by_id().get(doc_id) if self._blocks else None
        if row is None:
            raise KeyError(doc_id)
        return self._make(self._block_of(row), row, doc_id)

    def get(self, doc_id, default=None):
        try:
            return self[doc_id]
        except KeyError:
            return default

    def __contains__(self, doc_id) -> bool:
        return dict.__contains__(self, doc_id) or (bool(self._blocks) and doc_id in self._rows_by_id())

    def __len__(self) -> int:
        return dict.__len__(self) + self._lazy_count

    def materialize_all(self) -> None:
        if self._lazy_count:
            for j in range(len(self._blocks)):
                if self._block_on_device(j):
                    self._block_to_host(j)        # a whole-store view: the block's columns come over once
            for j, (row0, _, ids, _) in enumerate(self._blocks):
                made = self._made[j]
                for i, doc_id in enumerate(ids):
                    if row0 + i not in made and not dict.__contains__(self, doc_id):
                        self._make(j, row0 + i, doc_id)
            self._lazy_count = 0

    def __iter__(self):
        self.materialize_all()
        return dict.__iter__(self)

    def keys(self):
        self.materialize_all()
        return dict.keys(self)

    def values(self):
        self.materialize_all()
        return dict.values(self)

    def items(self):
        self.materialize_all()
        return dict.items(self)
