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
from typing import Dict, Iterable, Mapping, Sequence

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
