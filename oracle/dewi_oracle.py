"""CPU oracle for the DEWI scoring-and-retrieval hot path.

THIS FILE IS TEST INFRASTRUCTURE.  It is a NumPy restatement of the reference
algorithm, used ONLY as the checker by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  Nothing in the product package
(``dewi-design-…_amd/dewi``) imports it, and the product path raises when the
HIP extension is missing instead of falling back to this code.

Parity status: PINNED.  ``oracle/gen_golden.py`` runs the real reference
(imported from /root/reference/src in the build container) on seeded inputs and
stores its outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function below against those files bit for bit (ids, fp32 scores,
fp32 medians/MADs) or to 1 ulp-of-f64 (scorer output).

Every function cites the reference lines it restates (paths relative to
/root/reference).  The step order, dtypes and rounding sequence are the
reference's; only the container types differ (SoA arrays instead of per-doc
``Payload`` objects).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

SIGNAL_KEYS = ("ht_mean", "ht_q90", "hi_mean", "hi_q90", "I_hat", "redundancy", "noise")
PAYLOAD_KEYS = ("dewi",) + SIGNAL_KEYS


# ---------------------------------------------------------------------------
# A1 / A2 — ExactIndex.add + build                src/dewi/backends.py:394-412
# ---------------------------------------------------------------------------
def normalize_row(emb: np.ndarray, space: str = "cosine") -> np.ndarray:
    """One ``ExactIndex.add`` row: fp32 cast, then ``emb / ||emb||`` for cosine.

    backends.py:403-405.  No zero-norm guard: a zero row becomes NaN, exactly
    like the reference.
    """
    emb = np.asarray(emb).astype(np.float32)
    if space == "cosine":
        emb = emb / np.linalg.norm(emb)
    return emb


def build_matrix(rows: np.ndarray, space: str = "cosine") -> np.ndarray:
    """``add`` x N followed by ``build``: N x d C-order fp32 (backends.py:408-412)."""
    rows = np.asarray(rows)
    if rows.shape[0] == 0:
        raise ValueError("No embeddings to build index from")  # backends.py:409-410
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.stack([normalize_row(r, space) for r in rows])


def payload_soa(dewi: Sequence[float], ht_mean: Sequence[float], hi_mean: Sequence[float]
                ) -> Tuple[np.ndarray, np.ndarray]:
    """The two per-doc fp32 values ``ExactIndex.search`` reads from a Payload.

    backends.py:450-458: ``dewi_scores[i] = payload.dewi`` (f64 -> fp32 store) and
    ``entropies[i] = (payload.ht_mean + payload.hi_mean) * 0.5`` (f64 arithmetic,
    then fp32 store).
    """
    d = np.asarray(dewi, dtype=np.float64).astype(np.float32)
    e = ((np.asarray(ht_mean, dtype=np.float64) + np.asarray(hi_mean, dtype=np.float64)) * 0.5
         ).astype(np.float32)
    return d, e


# ---------------------------------------------------------------------------
# A3 — ExactIndex.search steps 1-4               src/dewi/backends.py:420-447
# ---------------------------------------------------------------------------
def prepare_query(query: np.ndarray, space: str = "cosine") -> np.ndarray:
    """backends.py:420-424: fp32 cast; divide by the norm unless it is zero."""
    q = np.asarray(query, dtype=np.float32)
    if space == "cosine":
        n = np.linalg.norm(q)
        if n > 0:
            q = q / n
    return q


def similarities(E: np.ndarray, q: np.ndarray, space: str = "cosine") -> np.ndarray:
    """backends.py:431-436: ``E @ q`` (cosine) or ``-sum((E - q)^2)`` (l2), fp32."""
    q2 = q.reshape(1, -1)
    if space == "cosine":
        return np.dot(E, q2.T).flatten()
    return -np.sum((E - q2) ** 2, axis=1)


def candidate_cut(scores: np.ndarray, k: int) -> np.ndarray:
    """backends.py:439-444: the top-min(2k, N) rows by similarity, unordered."""
    c = min(2 * k, len(scores))
    if c <= 0:
        return np.empty(0, dtype=np.int64)
    return np.argpartition(scores, -c)[-c:]


# ---------------------------------------------------------------------------
# A4 — ExactIndex.search steps 5-9               src/dewi/backends.py:450-481
# ---------------------------------------------------------------------------
def rerank(cand_idx: np.ndarray, cand_scores: np.ndarray, dewi32: np.ndarray, ent32: np.ndarray,
           k: int, eta: float, entropy_pref: float) -> Tuple[np.ndarray, np.ndarray]:
    """Blend, top-k, order.  Returns (row indices int64[k], adjusted fp32[k]).

    backends.py:461  ``(1 - eta) * s + eta * dewi`` — two rounded fp32 products
    and one fp32 add (eta is a Python float, hence a weak scalar).
    backends.py:464-465  ``+= entropy_pref * ent`` only when entropy_pref != 0.
    backends.py:468-471  argpartition top-k, then argsort of the negated scores.
    """
    adjusted = (1 - eta) * cand_scores + eta * dewi32[cand_idx]
    if entropy_pref != 0:
        adjusted += entropy_pref * ent32[cand_idx]
    top_k = np.argpartition(adjusted, -k)[-k:]  # raises ValueError when k > c, as the reference
    order = top_k[np.argsort(-adjusted[top_k])]
    return cand_idx[order].astype(np.int64), adjusted[order].astype(np.float32)


def search(E: np.ndarray, query: np.ndarray, dewi32: np.ndarray, ent32: np.ndarray, k: int = 10,
           eta: float = 0.5, entropy_pref: float = 0.0, space: str = "cosine",
           return_candidates: bool = False):
    """Whole ``ExactIndex.search`` (backends.py:414-481) on SoA payload arrays."""
    q = prepare_query(query, space)
    scores = similarities(E, q, space)
    cand = candidate_cut(scores, k)
    if cand.size == 0:
        out = (np.empty(0, np.int64), np.empty(0, np.float32))
        return out + (cand, scores[cand]) if return_candidates else out
    ids, adj = rerank(cand, scores[cand], dewi32, ent32, k, eta, entropy_pref)
    if return_candidates:
        return ids, adj, cand, scores[cand]
    return ids, adj


def search_prepared(E: np.ndarray, q_prepared: np.ndarray, dewi32: np.ndarray, ent32: np.ndarray, k: int,
                    eta: float, entropy_pref: float = 0.0, space: str = "cosine"):
    """``search`` without the query-normalisation step (the query is used as given)."""
    scores = similarities(E, np.asarray(q_prepared, dtype=np.float32), space)
    cand = candidate_cut(scores, k)
    if cand.size == 0:
        return np.empty(0, np.int64), np.empty(0, np.float32)
    return rerank(cand, scores[cand], dewi32, ent32, k, eta, entropy_pref)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bfloat16, returned as the fp32 values bf16 can hold (NaN kept).

    Config C3 stores the normalised corpus and the normalised query in bf16; products of two bf16
    values are exact in fp32, so running the fp32 restatement on rounded inputs IS the bf16 oracle.
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    out[nan] = x[nan]
    return out.reshape(x.shape)


# ---------------------------------------------------------------------------
# f64 companion used by the parity harness to decide which queries are
# "decisive" (SURVEY.md §8(a) note 5): fp32 summation order differs between
# OpenBLAS and any GPU reduction, so id-order equality is asserted only where
# every relevant gap exceeds a noise threshold.
# ---------------------------------------------------------------------------
def decision_gaps(E: np.ndarray, query: np.ndarray, dewi32: np.ndarray, ent32: np.ndarray, k: int,
                  eta: float, entropy_pref: float = 0.0, space: str = "cosine", exact: bool = True
                  ) -> Tuple[float, float]:
    """(similarity gap between rank c and c+1, smallest adjacent adjusted-score
    gap among ranks 1..k+1), computed in float64 from the fp32 inputs.  ``exact=False``
    takes the similarities from the fp32 product instead (for corpora where an f64 copy of
    the matrix is too expensive); the gaps then carry ~1e-7 of noise themselves."""
    prepared = space == "prepared"      # cosine scores of a query that is used as given
    if prepared:
        space = "cosine"
    qp = np.asarray(query, dtype=np.float32) if prepared else prepare_query(query, space)
    if exact:
        q = qp.astype(np.float64)
        E64 = E.astype(np.float64)
        s = E64 @ q if space == "cosine" else -np.sum((E64 - q[None, :]) ** 2, axis=1)
    else:
        s = similarities(E, qp, space).astype(np.float64)
    n = s.shape[0]
    c = min(2 * k, n)
    order = np.argsort(-s, kind="stable")
    cut_gap = np.inf if c >= n else float(s[order[c - 1]] - s[order[c]])
    cand = order[:c]
    adj = np.float64(np.float32(1 - eta)) * s[cand] + np.float64(np.float32(eta)) * dewi32[cand].astype(np.float64)
    if entropy_pref != 0:
        adj = adj + np.float64(np.float32(entropy_pref)) * ent32[cand].astype(np.float64)
    a = np.sort(adj)[::-1]
    top = a[: min(k + 1, a.shape[0])]
    rank_gap = np.inf if top.shape[0] < 2 else float(np.min(top[:-1] - top[1:]))
    return cut_gap, rank_gap


# ---------------------------------------------------------------------------
# A10 — ANN re-rank rule (HNSWIndex / FAISSIndex.search)
#                                               src/dewi/backends.py:204-241, 309-356
# ---------------------------------------------------------------------------
def ann_rerank(neigh_idx: np.ndarray, neigh_sim: np.ndarray, dewi: np.ndarray, ht_mean: np.ndarray,
               hi_mean: np.ndarray, eta: float, entropy_pref: float) -> Tuple[np.ndarray, np.ndarray]:
    """k neighbours (already converted to a similarity: ``1 - dist`` for hnswlib
    cosine, raw IP for faiss-IP, ``1 / (1 + dist)`` for faiss-L2) -> stable
    descending sort of ``(1-eta)*sim + eta*dewi [+ pref*(ht_mean+hi_mean)/2]``.
    Restated by reading only: hnswlib / faiss are not installed (parity unpinned)."""
    adj = (1 - eta) * neigh_sim + eta * dewi[neigh_idx]
    if entropy_pref != 0:
        adj = adj + entropy_pref * ((ht_mean[neigh_idx] + hi_mean[neigh_idx]) / 2)
    order = np.argsort(-adj, kind="stable")
    return neigh_idx[order], adj[order]


def ann_library_distance(E: np.ndarray, q_prepared: np.ndarray, space: str = "cosine") -> np.ndarray:
    """The distance an ANN library reports for each row, as the reference's backends consume it: hnswlib's
    cosine / inner-product spaces return ``1.0f - <e, q>`` (fp32); hnswlib ``l2`` and faiss ``METRIC_L2`` return
    the squared L2 distance.  (faiss ``METRIC_INNER_PRODUCT`` returns the inner product itself, which
    backends.py:335-336 uses as the score directly.)  Restated from the libraries' documented behaviour:
    hnswlib >=0.7,<0.8 and faiss-cpu >=1.7 (pyproject.toml:60-64) are not installed — parity unpinned."""
    s = similarities(E, np.asarray(q_prepared, dtype=np.float32), space)
    return (np.float32(1.0) - s).astype(np.float32) if space == "cosine" else (-s).astype(np.float32)


def ann_similarity(dist: np.ndarray, kind: str) -> np.ndarray:
    """Neighbour distance -> the similarity the reference blends.  ``one_minus_dist``: backends.py:229-231
    ``(1 - dist)`` (HNSWIndex); ``inv_one_plus_dist``: backends.py:337-338 ``1.0 / (1.0 + dist)`` (FAISSIndex
    with METRIC_L2).  ``dist`` is an np.float32 SCALAR in the reference and ``payload.dewi`` / ``eta`` are Python
    floats.  The reference pins numpy<2.0 (pyproject.toml:39): there value-based casting makes ``1.0/(1.0+dist)``,
    ``(1-eta)*(1-dist)`` and ``eta*payload.dewi`` float64 scalar arithmetic (backends.py:229-231, 337-346); under
    NumPy >= 2 (NEP 50, what this container runs) the same expressions stay fp32.  This restatement and the device blend
    (csrc/select_rerank.hip ``blend``) are the fp32 form, one rounding per operation, on the fp32 ``dewi32`` column:
    against the numpy<2 float64 form that is a deliberate deviation of at most ~1e-7 relative in the adjusted score
    (well inside north_star's 1e-5).  Parity of A10 is unpinned either way: hnswlib / faiss are not importable."""
    d = np.asarray(dist, dtype=np.float32)
    if kind == "one_minus_dist":
        return (np.float32(1.0) - d).astype(np.float32)
    if kind == "inv_one_plus_dist":
        return (np.float32(1.0) / (np.float32(1.0) + d)).astype(np.float32)
    raise ValueError(f"unknown similarity transform {kind!r}")


# ---------------------------------------------------------------------------
# A6 — scorer.RobustStats.fit                     src/dewi/scorer.py:18-26
# ---------------------------------------------------------------------------
def robust_fit(columns: Dict[str, np.ndarray]) -> Tuple[Dict[str, float], Dict[str, float]]:
    """Per key: fp32 array, fp32 median, fp32 median of |x - med|, ``or 1e-8``."""
    med: Dict[str, float] = {}
    mad: Dict[str, float] = {}
    for key, col in columns.items():
        v = np.asarray(col, dtype=np.float32)
        med[key] = float(np.median(v))
        mad[key] = float(np.median(np.abs(v - med[key]))) or 1e-8
    return med, mad


# ---------------------------------------------------------------------------
# A7 / A8 — RobustStats.z, DewiScorer._components, score, score_conditional
#                                               src/dewi/scorer.py:28-31, 49-89
# ---------------------------------------------------------------------------
def _z(val: np.ndarray, med: float, mad: float) -> np.ndarray:
    return (val - med) / (1.4826 * mad)  # scorer.py:28-31, float64 throughout


def score(columns: Dict[str, np.ndarray], med: Dict[str, float], mad: Dict[str, float],
          weights: Optional[Sequence[float]] = None, delta: float = 3.0,
          mode: str = "standard") -> np.ndarray:
    """Vectorised float64 restatement of ``DewiScorer.score`` / ``score_conditional``.

    ``weights`` = (alpha_t, alpha_i, alpha_m, alpha_r, alpha_n), default all 1.0
    (types.py:42-51).  Operation order follows scorer.py:49-58, 64-75, 77-89
    term by term so that every intermediate rounds as in the reference.
    """
    at, ai, am, ar, an = (1.0,) * 5 if weights is None else [float(w) for w in weights]
    c = {k: np.asarray(columns[k], dtype=np.float64) for k in SIGNAL_KEYS}
    Ht = 0.5 * (_z(c["ht_mean"], med["ht_mean"], mad["ht_mean"]) + _z(c["ht_q90"], med["ht_q90"], mad["ht_q90"]))
    Hi = 0.5 * (_z(c["hi_mean"], med["hi_mean"], mad["hi_mean"]) + _z(c["hi_q90"], med["hi_q90"], mad["hi_q90"]))
    I = _z(c["I_hat"], med["I_hat"], mad["I_hat"])
    R = _z(c["redundancy"], med["redundancy"], mad["redundancy"])
    Nz = _z(c["noise"], med["noise"], mad["noise"])
    if mode == "standard":
        U = at * Ht + ai * Hi - am * I - ar * R - an * Nz          # scorer.py:67-73
    elif mode == "conditional":
        U = at * (Ht - I) + ai * (Hi - I) - ar * R - an * Nz        # scorer.py:80-87
    else:
        raise ValueError(f"unknown mode {mode!r}")
    U = np.clip(U, -delta, delta)                                     # scorer.py:74, 88
    return 1.0 / (1.0 + np.exp(-U))                                   # scorer.py:60-62


# ---------------------------------------------------------------------------
# Synthetic inputs — the distributions of the reference's own harness
#                        scripts/profile_index.py:49-70, tests/test_index.py:31-50
# Drawn as vectorised arrays per field; this is the generator
# bench.py and the parity tests share, so both sides always see the same data.
# ---------------------------------------------------------------------------
def synth_corpus(n: int, dim: int, seed: int = 42, chunk: int = 65536) -> np.ndarray:
    """Unit-norm gaussian rows, fp32, generated in chunks to bound memory."""
    rng = np.random.RandomState(seed)
    out = np.empty((n, dim), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        blk = rng.randn(e - s, dim).astype(np.float32)
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)
        out[s:e] = blk
    return out


def synth_payload_columns(n: int, seed: int = 42) -> Dict[str, np.ndarray]:
    """The 8 payload fields as float64 columns, order as listed in SURVEY §8(d)."""
    rng = np.random.RandomState(seed + 1000)
    return {
        "dewi": np.clip(rng.beta(2, 2, n), 0, 1),
        "ht_mean": rng.gamma(2, 0.5, n),
        "ht_q90": rng.gamma(2, 0.5, n) * 1.5,
        "hi_mean": rng.gamma(2, 0.3, n),
        "hi_q90": rng.gamma(2, 0.3, n) * 1.5,
        "I_hat": rng.beta(2, 2, n),
        "redundancy": rng.beta(1, 5, n),
        "noise": rng.beta(1, 10, n),
    }


def synth_queries(n_queries: int, dim: int, seed: int = 7) -> np.ndarray:
    return np.random.RandomState(seed).randn(n_queries, dim).astype(np.float32)
