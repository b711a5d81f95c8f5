"""Parity harness shared by the GPU tests: HIP result vs the CPU oracle.

Rule (SURVEY.md §8(a) note 5): fp32 dot products are summed in a different order by OpenBLAS
and by the GPU (~1e-7 relative), so a query is compared in one of two ways.

* DECISIVE query — the similarity gap at the candidate cut and the smallest adjacent
  adjusted-score gap among ranks 1..k+1 (both computed in float64) exceed GAP (5e-7, scaled by
  the score magnitude for l2): the id ranking must EQUAL the oracle's, scores within 1e-5.
* near-tie query — an admissible-outcome check that still compares ids, not just scores:
  (a) every returned row is an admissible candidate (its f64 similarity is not below the c-th
      best by more than GAP),
  (b) every returned score equals the oracle's fp32 blend FOR THAT ROW within the tolerance
      (this pins the id <-> score pairing),
  (c) no row that is surely a candidate (similarity above the first excluded row by more than
      GAP) and beats the worst returned adjusted score by more than GAP was left out,
  (d) no duplicates, scores non-increasing.
  So the returned id SET can differ from the oracle's only in rows within GAP of a decision
  boundary, and the order only between rows whose adjusted scores are within the tolerance.

``check_batch`` returns the decisive count and asserts a floor on it (default: 80 % of the
queries at k <= 10, 25 % above): no case can pass on near-tie checks alone.
"""
import numpy as np

import dewi_oracle as orc

GAP = 5e-7
SCORE_TOL = 1e-5


def _reference(E, q, dewi32, ent32, k, eta, pref, space, exact_gaps, prepared):
    """Oracle result + the float64 companion quantities the near-tie rules need."""
    qp = np.asarray(q, dtype=np.float32) if prepared else orc.prepare_query(q, space)
    s32 = orc.similarities(E, qp, space)                                   # backends.py:431-436
    cand = orc.candidate_cut(s32, k)
    if cand.size:
        ref_ids, ref_sc = orc.rerank(cand, s32[cand], dewi32, ent32, k, eta, pref)
    else:
        ref_ids, ref_sc = np.empty(0, np.int64), np.empty(0, np.float32)
    if exact_gaps:
        q64 = qp.astype(np.float64)
        E64 = E.astype(np.float64)
        s64 = E64 @ q64 if space != "l2" else -np.sum((E64 - q64[None, :]) ** 2, axis=1)
    else:
        s64 = s32.astype(np.float64)
    w_sim, w_dewi, w_ent = np.float64(np.float32(1 - eta)), np.float64(np.float32(eta)), np.float64(np.float32(pref))
    adj64 = w_sim * s64 + w_dewi * dewi32.astype(np.float64)
    if pref != 0:
        adj64 = adj64 + w_ent * ent32.astype(np.float64)
    return s32, s64, adj64, ref_ids, ref_sc


def _row_scores32(s32, rows, dewi32, ent32, eta, pref):
    """The oracle's fp32 blend (backends.py:461-465) for given rows."""
    adj = (1 - eta) * s32[rows] + eta * dewi32[rows]
    if pref != 0:
        adj += pref * ent32[rows]
    return adj.astype(np.float32)


def compare_query(E, q, dewi32, ent32, k, eta, pref, space, got_ids, got_scores, exact_gaps=True, gap=GAP,
                  score_tol=SCORE_TOL, prepared=False):
    """Returns (decisive: bool, message or None).  ``prepared``: q is used as given (no normalisation)."""
    s32, s64, adj64, ref_ids, ref_sc = _reference(E, q, dewi32, ent32, k, eta, pref, space, exact_gaps, prepared)
    n = s64.shape[0]
    c = min(2 * k, n)
    got_ids = np.asarray(got_ids)
    got_scores = np.asarray(got_scores)
    if got_ids.shape != ref_ids.shape:
        return False, f"shape {got_ids.shape} != {ref_ids.shape}"
    if c <= 0 or ref_ids.size == 0:
        return True, None
    order = np.argsort(-s64, kind="stable")
    v_c = s64[order[c - 1]]
    v_c1 = s64[order[c]] if c < n else -np.inf
    cut_gap = v_c - v_c1
    top = np.sort(adj64[order[:c]])[::-1][: min(k + 1, c)]
    rank_gap = np.inf if top.shape[0] < 2 else float(np.min(top[:-1] - top[1:]))
    # Noise scales with the magnitude of what is compared: the similarity cut with |similarity at the cut| (l2 scores are
    # -||e - q||^2, hundreds; at eta = 1 the ADJUSTED scores stay O(1) while the cut still happens among those hundreds —
    # found by scripts/fuzz_parity.py), the ranking with the adjusted scores (which carry w_sim times the similarity).
    sim_scale = max(1.0, float(abs(v_c))) if np.isfinite(v_c) else 1.0
    scale = max(1.0, float(np.max(np.abs(ref_sc))), float(abs(np.float32(1 - eta))) * sim_scale)
    g_cut = gap * sim_scale
    g = gap * scale
    decisive = bool(cut_gap > g_cut and rank_gap > g)
    if np.isnan(s64).any():
        # NaN rows (zero-norm embeddings) rank first as in NumPy: order among them is an artefact, so only
        # the exact comparison is meaningful and only when nothing else is near a tie
        decisive = False
    if got_scores.size > 1 and not np.all((got_scores[:-1] >= got_scores[1:]) | np.isnan(got_scores[:-1])):
        return decisive, "scores not non-increasing"
    if decisive:
        if not np.array_equal(got_ids, ref_ids):
            return decisive, f"ids differ: got {got_ids.tolist()} want {ref_ids.tolist()} (gaps {cut_gap:.2e}, {rank_gap:.2e})"
        err = float(np.max(np.abs(got_scores.astype(np.float64) - ref_sc.astype(np.float64))))
        if not err <= score_tol * scale:
            return decisive, f"score error {err:.3e} > {score_tol * scale:.1e}"
        return decisive, None
    # ---- near-tie query: admissible-outcome check on the ids
    gi = got_ids.astype(np.int64)
    if gi.min() < 0 or gi.max() >= n:
        return decisive, f"row index out of range: {gi.min()}..{gi.max()}"
    if len(set(gi.tolist())) != gi.size:
        return decisive, "duplicate rows in the result"
    nan_rows = np.isnan(s64)
    ok_rows = ~nan_rows[gi]
    if np.any(s64[gi][ok_rows] < v_c - g_cut) and not np.isnan(v_c):
        bad = gi[ok_rows][s64[gi][ok_rows] < v_c - g_cut]
        return decisive, f"rows {bad.tolist()} are not among the top-{c} similarities (cut {v_c:.7f}, gap {g_cut:.1e})"
    want_sc = _row_scores32(s32, gi, dewi32, ent32, eta, pref)
    with np.errstate(invalid="ignore"):
        err = np.abs(got_scores.astype(np.float64) - want_sc.astype(np.float64))
    err = err[~np.isnan(want_sc)]
    if err.size and not float(np.max(err)) <= score_tol * scale:
        return decisive, f"near-tie query: score of a returned row off by {float(np.max(err)):.3e}"
    if not nan_rows.any():
        sure = s64 > v_c1 + g_cut if c < n else np.ones(n, dtype=bool)
        sure[gi] = False
        worst = float(np.min(adj64[gi]))
        if np.any(adj64[sure] > worst + g):
            x = np.nonzero(sure)[0][int(np.argmax(adj64[sure]))]
            return decisive, (f"near-tie query: row {int(x)} (adjusted {adj64[x]:.7f}) is a sure candidate and beats the "
                              f"worst returned score {worst:.7f} by more than {g:.1e} but was left out")
    return decisive, None


def device_prepared_queries(Q, space="cosine"):
    """GPU only: the library's own prepared bf16 queries (``dewi_prepare_queries_bf16``; every bf16 kernel
    normalises with the same float64-summed norm, see csrc/common.hpp ``wave_query_norm``) as fp32 values,
    AFTER checking them against the oracle's ``bf16_round(prepare_query(q))``: every element within one bf16
    ulp, at least 99 % bit-equal.  (NumPy sums ||q||^2 in fp32, so an element on a rounding boundary may
    round the other way; feeding the oracle these queries removes that one-ulp noise from the comparison,
    which can then be made at fp32-summation tolerance.)"""
    import torch
    from dewi import _engine as eng
    qd = torch.from_numpy(np.ascontiguousarray(Q, dtype=np.float32)).cuda()
    got = eng.prepare_queries_bf16(qd, space).float().cpu().numpy()
    want = np.stack([orc.bf16_round(orc.prepare_query(q, space)) for q in np.atleast_2d(Q)])
    ulp = np.abs(want) * 2.0 ** -7 + 1e-38            # one bf16 ulp is at most 2^-7 relative
    assert np.all(np.abs(got - want) <= ulp), "a prepared query element is more than one bf16 ulp off the oracle's"
    assert np.mean(got == want) >= 0.99, float(np.mean(got == want))
    return got


def default_floor(k: int) -> float:
    return 0.8 if k <= 10 else 0.25


def check_batch(E, Q, dewi32, ent32, k, eta, pref, space, ids, scores, min_decisive_frac=None, **kw):
    """Every query must pass ``compare_query``; at least ``min_decisive_frac`` of them (default by k, see
    the module docstring; never zero) must have been decisive, i.e. compared id for id.  Returns the
    decisive count."""
    floor = default_floor(k) if min_decisive_frac is None else float(min_decisive_frac)
    assert floor > 0.0, "a parity case must assert some decisive id comparisons"
    n_dec = 0
    for j in range(Q.shape[0]):
        decisive, msg = compare_query(E, Q[j], dewi32, ent32, k, eta, pref, space, ids[j], scores[j], **kw)
        assert msg is None, f"query {j}: {msg}"
        n_dec += 1 if decisive else 0
    need = int(np.ceil(floor * Q.shape[0]))
    assert n_dec >= need, f"only {n_dec}/{Q.shape[0]} queries were decisive (floor {need}): pick other seeds or a tighter gap"
    return n_dec


def count_decisive(E, Q, dewi32, ent32, k, eta, pref, space, exact_gaps=True, gap=GAP, prepared=False):
    """CPU-only: how many queries of a case are decisive (used to calibrate the floors)."""
    n_dec = 0
    for j in range(Q.shape[0]):
        _, s64, adj64, ref_ids, ref_sc = _reference(E, Q[j], dewi32, ent32, k, eta, pref, space, exact_gaps, prepared)
        n = s64.shape[0]
        c = min(2 * k, n)
        order = np.argsort(-s64, kind="stable")
        cut_gap = s64[order[c - 1]] - (s64[order[c]] if c < n else -np.inf)
        top = np.sort(adj64[order[:c]])[::-1][: min(k + 1, c)]
        rank_gap = np.inf if top.shape[0] < 2 else float(np.min(top[:-1] - top[1:]))
        v_c = s64[order[c - 1]]
        sim_scale = max(1.0, float(abs(v_c))) if np.isfinite(v_c) else 1.0
        scale = max(1.0, float(np.max(np.abs(ref_sc))) if ref_sc.size else 1.0, float(abs(np.float32(1 - eta))) * sim_scale)
        n_dec += 1 if (cut_gap > gap * sim_scale and rank_gap > gap * scale) and not np.isnan(s64).any() else 0
    return n_dec
