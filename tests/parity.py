"""Parity harness shared by the GPU tests: HIP result vs the CPU oracle.

Rule (SURVEY.md §8(a) note 5): fp32 dot products are summed in a different order by
OpenBLAS and by the GPU (~1e-7 relative), so the id ranking is compared exactly only
for "decisive" queries — those whose similarity gap at the candidate cut and whose
smallest adjacent adjusted-score gap in the top k+1 both exceed GAP (5e-7, scaled by the
score magnitude for l2).  Scores must agree to 1e-5 (relative for |score| > 1).  The
number of non-decisive queries is returned so tests can bound it.
"""
import numpy as np

import dewi_oracle as orc

GAP = 5e-7
SCORE_TOL = 1e-5


def compare_query(E, q, dewi32, ent32, k, eta, pref, space, got_ids, got_scores, exact_gaps=True, gap=GAP,
                  score_tol=SCORE_TOL, prepared=False):
    """Returns (decisive: bool, message or None).  ``prepared``: q is used as given (no normalisation)."""
    if prepared:
        ref_ids, ref_sc = orc.search_prepared(E, q, dewi32, ent32, k, eta, pref, space)
        cut_gap, rank_gap = orc.decision_gaps(E, q, dewi32, ent32, k, eta, pref, "l2" if space == "l2" else "prepared",
                                              exact=exact_gaps)
    else:
        ref_ids, ref_sc = orc.search(E, q, dewi32, ent32, k, eta, pref, space)
        cut_gap, rank_gap = orc.decision_gaps(E, q, dewi32, ent32, k, eta, pref, space, exact=exact_gaps)
    scale = max(1.0, float(np.max(np.abs(ref_sc))) if ref_sc.size else 1.0)
    decisive = min(cut_gap, rank_gap) > gap * scale
    SCORE_TOL_ = score_tol
    got_ids = np.asarray(got_ids)
    got_scores = np.asarray(got_scores)
    if got_ids.shape != ref_ids.shape:
        return decisive, f"shape {got_ids.shape} != {ref_ids.shape}"
    if decisive:
        if not np.array_equal(got_ids, ref_ids):
            return decisive, f"ids differ: got {got_ids.tolist()} want {ref_ids.tolist()} (gaps {cut_gap:.2e}, {rank_gap:.2e})"
        err = np.max(np.abs(got_scores.astype(np.float64) - ref_sc.astype(np.float64))) if ref_sc.size else 0.0
        if not err <= SCORE_TOL_ * scale:
            return decisive, f"score error {err:.3e} > {SCORE_TOL_ * scale:.1e}"
    else:
        # near-tie: same scores in sorted order within tolerance is all that can be asked
        a = np.sort(got_scores.astype(np.float64))
        b = np.sort(ref_sc.astype(np.float64))
        if a.shape == b.shape and a.size and np.max(np.abs(a - b)) > 10 * SCORE_TOL_ * scale:
            return decisive, "near-tie query: score multiset differs"
    if got_scores.size > 1 and not np.all(got_scores[:-1] >= got_scores[1:]):
        return decisive, "scores not non-increasing"
    return decisive, None


def check_batch(E, Q, dewi32, ent32, k, eta, pref, space, ids, scores, max_excluded_frac=0.05, **kw):
    excluded = 0
    for j in range(Q.shape[0]):
        decisive, msg = compare_query(E, Q[j], dewi32, ent32, k, eta, pref, space, ids[j], scores[j], **kw)
        assert msg is None, f"query {j}: {msg}"
        excluded += 0 if decisive else 1
    assert excluded <= max(1, int(max_excluded_frac * Q.shape[0])), f"{excluded}/{Q.shape[0]} queries were near-ties"
    return excluded
