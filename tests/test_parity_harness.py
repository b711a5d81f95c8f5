"""CPU self-test of tests/parity.py: the harness must accept the oracle's own answers and REJECT
wrong ones — on decisive queries by exact id comparison, on near-tie queries by the id-set rules
(a wrong row, a wrong score for a row, an omitted sure candidate), never by score multisets alone.
"""
import numpy as np
import pytest

import dewi_oracle as orc
import parity


def _case(n=3000, dim=48, b=12, seed=5):
    raw = orc.synth_corpus(n, dim, seed=seed)
    cols = orc.synth_payload_columns(n, seed=seed)
    E = orc.build_matrix(raw)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = orc.synth_queries(b, dim, seed=seed + 1)
    return E, dewi32, ent32, Q


def _answers(E, Q, dewi32, ent32, k, eta, pref):
    out = [orc.search(E, q, dewi32, ent32, k, eta, pref) for q in Q]
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])


def test_oracle_answers_pass_and_floor_is_enforced():
    E, dewi32, ent32, Q = _case()
    ids, sc = _answers(E, Q, dewi32, ent32, 10, 0.3, 0.1)
    n_dec = parity.check_batch(E, Q, dewi32, ent32, 10, 0.3, 0.1, "cosine", ids, sc)
    assert n_dec >= 10
    assert n_dec == parity.count_decisive(E, Q, dewi32, ent32, 10, 0.3, 0.1, "cosine")
    with pytest.raises(AssertionError, match="decisive"):
        # an absurd gap makes every query a near-tie: the floor must then fail the case
        parity.check_batch(E, Q, dewi32, ent32, 10, 0.3, 0.1, "cosine", ids, sc, gap=1.0)
    with pytest.raises(AssertionError, match="must assert some decisive"):
        parity.check_batch(E, Q, dewi32, ent32, 10, 0.3, 0.1, "cosine", ids, sc, min_decisive_frac=0.0)


def test_decisive_query_rejects_swapped_ids_and_score_drift():
    E, dewi32, ent32, Q = _case()
    ids, sc = _answers(E, Q, dewi32, ent32, 10, 0.3, 0.0)
    dec, msg = parity.compare_query(E, Q[0], dewi32, ent32, 10, 0.3, 0.0, "cosine", ids[0], sc[0])
    assert dec and msg is None
    swapped = ids[0].copy()
    swapped[[3, 4]] = swapped[[4, 3]]
    assert "ids differ" in parity.compare_query(E, Q[0], dewi32, ent32, 10, 0.3, 0.0, "cosine", swapped, sc[0])[1]
    drift = sc[0].copy()
    drift[2] += 3e-5
    assert "score error" in parity.compare_query(E, Q[0], dewi32, ent32, 10, 0.3, 0.0, "cosine", ids[0], drift)[1]


def test_near_tie_query_still_compares_ids():
    """Force the near-tie branch with a large gap, then feed wrong answers whose score MULTISET is right or
    nearly right: the old harness accepted these, the id-set rules must not."""
    E, dewi32, ent32, Q = _case()
    k, eta, pref = 10, 0.3, 0.0
    ids, sc = _answers(E, Q, dewi32, ent32, k, eta, pref)
    kw = dict(gap=2e-3)                                   # > the typical adjusted-score spacing: nothing is decisive
    dec, msg = parity.compare_query(E, Q[1], dewi32, ent32, k, eta, pref, "cosine", ids[1], sc[1], **kw)
    assert not dec and msg is None
    # (b) a row that is NOT in the result, carrying the score of the row it replaces
    s = orc.similarities(E, orc.prepare_query(Q[1]))
    outsider = int(np.argsort(s)[len(s) // 2])           # a mid-ranked row: far below the candidate cut
    wrong = ids[1].copy()
    wrong[5] = outsider
    m = parity.compare_query(E, Q[1], dewi32, ent32, k, eta, pref, "cosine", wrong, sc[1], **kw)[1]
    assert m is not None and "not among the top" in m
    # (b') an admissible candidate row, but with another row's score attached
    cand = orc.candidate_cut(s, k)
    spare = [int(r) for r in cand if r not in set(ids[1].tolist())]
    wrong = ids[1].copy()
    wrong[9] = spare[0]
    adj_spare = parity._row_scores32(s, np.array([spare[0]]), dewi32, ent32, eta, pref)[0]
    if abs(float(adj_spare) - float(sc[1][9])) > 1e-4:
        m = parity.compare_query(E, Q[1], dewi32, ent32, k, eta, pref, "cosine", wrong, sc[1], **kw)[1]
        assert m is not None and "score of a returned row" in m
    # (c) the best row dropped, everything shifted up, a far worse admissible candidate appended with ITS score
    worst_first = sorted(spare, key=lambda r: float(parity._row_scores32(s, np.array([r]), dewi32, ent32, eta, pref)[0]))
    tail = worst_first[0]
    tail_sc = parity._row_scores32(s, np.array([tail]), dewi32, ent32, eta, pref)[0]
    if float(sc[1][0]) - float(tail_sc) > 3e-3:
        wrong = np.concatenate([ids[1][1:], [tail]])
        wsc = np.concatenate([sc[1][1:], [tail_sc]]).astype(np.float32)
        m = parity.compare_query(E, Q[1], dewi32, ent32, k, eta, pref, "cosine", wrong, wsc, **kw)[1]
        assert m is not None and "left out" in m
    # (d) duplicates
    dup = ids[1].copy()
    dup[1] = dup[0]
    dsc = sc[1].copy()
    dsc[1] = dsc[0]
    assert "duplicate" in parity.compare_query(E, Q[1], dewi32, ent32, k, eta, pref, "cosine", dup, dsc, **kw)[1]


def test_near_tie_accepts_a_legitimate_swap():
    """Two rows whose adjusted scores differ by less than the gap may come back in either order."""
    E, dewi32, ent32, Q = _case()
    k, eta, pref = 10, 0.3, 0.0
    ids, sc = _answers(E, Q, dewi32, ent32, k, eta, pref)
    gaps = sc[2][:-1] - sc[2][1:]
    j = int(np.argmin(gaps))
    g = float(gaps[j]) * 1.5 + 1e-7
    sw, ssc = ids[2].copy(), sc[2].copy()
    sw[[j, j + 1]] = sw[[j + 1, j]]
    ssc[[j, j + 1]] = ssc[[j + 1, j]]
    ssc = np.sort(ssc)[::-1].copy()                        # still non-increasing; each score within g of its row's
    dec, msg = parity.compare_query(E, Q[2], dewi32, ent32, k, eta, pref, "cosine", sw, ssc, gap=g, score_tol=g)
    assert not dec and msg is None


def test_l2_and_prepared_modes():
    rs = np.random.RandomState(3)
    E = (rs.randn(500, 24) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(500, seed=3)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = (rs.randn(6, 24) * 0.5).astype(np.float32)
    out = [orc.search(E, q, dewi32, ent32, 7, 0.5, -0.2, "l2") for q in Q]
    ids, sc = np.stack([o[0] for o in out]), np.stack([o[1] for o in out])
    assert parity.check_batch(E, Q, dewi32, ent32, 7, 0.5, -0.2, "l2", ids, sc, min_decisive_frac=0.5) >= 3
    Eb = orc.bf16_round(orc.build_matrix(rs.randn(400, 32).astype(np.float32)))
    Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in rs.randn(5, 32).astype(np.float32)])
    out = [orc.search_prepared(Eb, q, dewi32[:400], ent32[:400], 5, 0.3, 0.0) for q in Qp]
    ids, sc = np.stack([o[0] for o in out]), np.stack([o[1] for o in out])
    parity.check_batch(Eb, Qp, dewi32[:400], ent32[:400], 5, 0.3, 0.0, "cosine", ids, sc, prepared=True, min_decisive_frac=0.5)
