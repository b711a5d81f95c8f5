"""GPU tests of the round-2 additions: thread safety of the blocking search, per-thread tuning,
column ingest with lazy payloads, the ANN-semantics similarity transforms (A10) and the scorer's
row API served from the fitted table.
"""
import threading
import time

import numpy as np
import pytest

import dewi_oracle as orc

pytestmark = pytest.mark.gpu


def _corpus(n, d, seed, space="cosine"):
    from dewi import _engine as eng
    raw = orc.synth_corpus(n, d, seed=seed) if space == "cosine" else \
        (np.random.RandomState(seed).randn(n, d) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=seed)
    return eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space), raw, cols


def test_two_threads_share_one_index_and_own_streams():
    """Thread A and thread B hammer the SAME DeviceCorpus through the blocking search (serialised by the
    per-corpus lock: staging buffers are shared) while each also drives its OWN corpus on its own stream.
    Every answer must equal the serial answer.  Thread B additionally switches the matrix-core path off with
    the thread-local tuning: that must not leak into thread A."""
    import torch
    from dewi import _engine as eng
    shared, _, _ = _corpus(40_000, 768, seed=61)
    Q = orc.synth_queries(24, 768, seed=62)
    want = [shared.search(Q[j], 10, 0.3, 0.1) for j in range(24)]
    own = []
    for t in range(2):
        cb, _, _ = _corpus(70_000, 256, seed=63 + t)
        own.append(cb.to_bf16())
    Qb = orc.synth_queries(16, 256, seed=65)
    own_want = [c.search(Qb, 10, 0.3, 0.0) for c in own]         # 16 queries: matrix-core path
    eng.tuning(0, 0, -1, 0)
    try:
        own_want_exact = [c.search(Qb, 10, 0.3, 0.0) for c in own]   # same, small-batch kernels
    finally:
        eng.tuning(0, 0, -1, 1)
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                if t == 1:
                    eng.tuning(0, 0, -1, 0)                       # this thread only
                for rep in range(6):
                    for j in range(t, 24, 2):
                        ids, sc = shared.search(Q[j], 10, 0.3, 0.1)
                        if not (np.array_equal(ids, want[j][0]) and np.array_equal(sc, want[j][1])):
                            errors.append(f"thread {t}: shared query {j} differs")
                    ids, sc = own[t].search(Qb, 10, 0.3, 0.0)
                    ref = own_want_exact[t] if t == 1 else own_want[t]
                    if not (np.array_equal(ids, ref[0]) and np.array_equal(sc, ref[1])):
                        errors.append(f"thread {t}: own corpus differs (tuning leaked?)")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    # the main thread's tuning is untouched by thread 1's override
    ids, sc = own[0].search(Qb, 10, 0.3, 0.0)
    assert np.array_equal(ids, own_want[0][0]) and np.array_equal(sc, own_want[0][1])


def test_add_batch_columns_equals_add_batch_and_is_lazy():
    from dewi.index import DewiIndex
    from dewi.types import payloads_from_columns
    n, d = 50_000, 128
    raw = orc.synth_corpus(n, d, seed=71)
    cols = orc.synth_payload_columns(n, seed=71)
    ids = [f"doc_{i:08d}" for i in range(n)]
    a = DewiIndex(dim=d, use_ann=False)
    a.add_batch(ids, raw, payloads_from_columns(cols))
    b = DewiIndex(dim=d, use_ann=False)
    t0 = time.perf_counter()
    b.add_batch_columns(ids[: n // 2], raw[: n // 2], {k: v[: n // 2] for k, v in cols.items()})
    b.add_batch_columns(ids[n // 2:], raw[n // 2:], {k: v[n // 2:] for k, v in cols.items()})
    ingest_s = time.perf_counter() - t0
    assert ingest_s < 0.5, ingest_s                               # no per-row Python work
    assert len(b) == n and dict.__len__(b._backend._payloads) == 0   # nothing materialised yet
    Q = orc.synth_queries(5, d, seed=72)
    for q in Q:
        ra, rb = a.search(q, k=10, eta=0.3, entropy_pref=0.2), b.search(q, k=10, eta=0.3, entropy_pref=0.2)
        assert [(r[0], r[1]) for r in ra] == [(r[0], r[1]) for r in rb]
        assert [r[2] for r in ra] == [r[2] for r in rb]           # equal Payload values
    made = dict.__len__(b._backend._payloads)
    assert 0 < made <= 50                                          # only the rows that were returned
    r1, r2 = b.search(Q[0], k=3), b.search(Q[0], k=3)
    assert all(x[2] is y[2] for x, y in zip(r1, r2))              # the same object every time
    assert b.get_payload(r1[0][0]) is r1[0][2]
    p = b.get_payload("doc_00000007")
    assert p.dewi == cols["dewi"][7] and p.noise == cols["noise"][7]
    assert b.get_payload("nope") is None
    # an in-place edit of a returned Payload is seen by refresh_payloads, as for object ingest
    top = b.search(Q[1], k=1, eta=1.0)[0]
    top[2].dewi = -5.0
    b._backend.refresh_payloads()
    assert b.search(Q[1], k=1, eta=1.0)[0][0] != top[0]
    with pytest.raises(ValueError, match="unknown payload columns"):
        b.add_batch_columns(["x"], raw[:1], {"bogus": np.zeros(1)})
    with pytest.raises(ValueError, match="shape"):
        b.add_batch_columns(["x"], raw[:1], {"dewi": np.zeros(2)})


@pytest.mark.parametrize("space,kind", [("cosine", "ip"), ("cosine", "one_minus_dist"), ("l2", "one_minus_dist"),
                                        ("l2", "inv_one_plus_dist"), ("cosine", "inv_one_plus_dist")])
def test_ann_similarity_transforms(space, kind):
    """A10: candidates=k with the similarity the reference's HNSW / FAISS backends blend
    (backends.py:229-231 `1 - dist`, :335-338 raw inner product / `1/(1+dist)`), on EXACT neighbours.
    PARITY UNPINNED: hnswlib / faiss are not installed here, so the rule is restated by reading
    (oracle.ann_library_distance / ann_similarity / ann_rerank), not checked against the libraries."""
    n, d, k, eta, pref = 4000, 128, 10, 0.4, 0.2
    c, raw, cols = _corpus(n, d, seed=81, space=space)
    E = c.emb.cpu().numpy()
    Q = orc.synth_queries(6, d, seed=82) * (0.5 if space == "l2" else 1.0)
    ids, sc = c.search(Q, k, eta, pref, candidates=k, similarity=kind)
    for j in range(Q.shape[0]):
        qp = orc.prepare_query(Q[j], space)
        s = orc.similarities(E, qp, space)
        nn = np.argsort(-s.astype(np.float64), kind="stable")[:k]
        if kind == "ip":
            sim = s[nn]
        else:
            sim = orc.ann_similarity(orc.ann_library_distance(E, qp, space)[nn], kind)
        want_ids, want_sc = orc.ann_rerank(nn, sim.astype(np.float64), cols["dewi"], cols["ht_mean"], cols["hi_mean"], eta, pref)
        assert sorted(ids[j].tolist()) == sorted(nn.tolist())
        assert np.array_equal(ids[j], want_ids), (j, ids[j], want_ids)
        assert np.allclose(sc[j], want_sc, rtol=0, atol=1e-5 * max(1.0, float(np.abs(want_sc).max())))
    with pytest.raises(ValueError, match="candidates=k"):
        c.search(Q, k, eta, pref, similarity="one_minus_dist")
    with pytest.raises(ValueError, match="unknown similarity"):
        c.search(Q, k, eta, pref, candidates=k, similarity="bogus")


def test_scorer_row_api_serves_the_fitted_table():
    """The reference's canonical caller (pipelines.py:208-221): fit_stats(signals) and then score(sig) for every
    dict of that list.  One kernel launch over the table, then rows from the cached column — bit for bit what
    the single-document launch returns — and a row edited after the fit is scored afresh."""
    from dewi.scorer import DewiScorer
    n = 100_000
    cols = orc.synth_payload_columns(n, seed=91)
    rows = [{key: float(cols[key][i]) for key in orc.SIGNAL_KEYS} for i in range(n)]
    sc = DewiScorer()
    sc.fit_stats(rows)
    t0 = time.perf_counter()
    got = np.array([sc.score(r) for r in rows])
    loop_s = time.perf_counter() - t0
    assert loop_s < 5.0, loop_s                                   # was ~0.1 ms of device round trip per row
    batch = sc.score_batch({key: cols[key] for key in orc.SIGNAL_KEYS})
    assert np.array_equal(got, batch)
    med, mad = orc.robust_fit({key: cols[key].astype(np.float32) for key in orc.SIGNAL_KEYS})
    ref = orc.score({key: cols[key] for key in orc.SIGNAL_KEYS}, med, mad)
    assert np.max(np.abs(got - ref) / ref) < 1e-15
    cond = np.array([sc.score_conditional(r) for r in rows[:2000]])
    refc = orc.score({key: cols[key][:2000] for key in orc.SIGNAL_KEYS}, med, mad, mode="conditional")
    assert np.max(np.abs(cond - refc) / refc) < 1e-15
    # out of order, and a dict that is not part of the table
    assert sc.score(rows[777]) == got[777] and sc.score(rows[5]) == got[5]
    clone = dict(rows[777])
    assert sc.score(clone) == got[777]                            # single-document launch: same bits
    e = int(np.argmin(np.abs(got - 0.5)))                         # a row in the middle of the sigmoid (not clipped)
    rows[e]["noise"] += 0.25                                      # edited after the fit: must not come from the cache
    fresh = sc.score(rows[e])
    assert fresh != got[e]
    assert fresh == sc.score(dict(rows[e]))
    # changing the weights invalidates the cached column
    m = int(np.argsort(np.abs(got - 0.5))[1])                     # another unclipped row
    sc.weights.alpha_n = 2.0
    assert sc.score(rows[m]) == sc.score(dict(rows[m])) != got[m]
