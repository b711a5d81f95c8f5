"""GPU tests added in round 3, all through the C ABI.

* the doc-id sharded path answers every k the single device answers: shard candidate records beyond 2048 per
  shard (dewi_knn_candidates / dewi_knn_scan + dewi_knn_finish sort in the workspace) and merges of more than
  2048 records per query (dewi_merge_rerank rank-merges the sorted shard lists through a workspace) — the
  reference has no such limit (backends.py:439-471);
* space="l2" batches keep the reference's arithmetic by default (the matrix-core form 2<e,q> - ||e||^2 - ||q||^2 is an
  opt-in): near-duplicate queries at large norms, unscaled 1e-5 tolerance, batch == single query bit for bit;
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import check_batch, compare_query

pytestmark = pytest.mark.gpu


def _engine():
    from dewi import _engine
    return _engine


def _shards(eng, raw, cols, bounds, space="cosine", bf16=False):
    out = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sub = {key: v[lo:hi] for key, v in cols.items()}
        sh = eng.DeviceCorpus.from_host(raw[lo:hi], sub["dewi"], sub["ht_mean"], sub["hi_mean"], space, id_offset=lo)
        out.append(sh.to_bf16() if bf16 else sh)
    return out


@pytest.mark.parametrize("n,bounds,k", [
    (9000, [0, 1000, 2300, 2310, 4000, 5500, 7000, 8000, 9000], 200),      # 8 shards x 400 records: 3200 > 2048 (LDS limit)
    (9000, [0, 1000, 2300, 2310, 4000, 5500, 7000, 8000, 9000], 1500),     # c = 3000: every shard is short of c
    (30000, [0, 9000, 19000, 30000], 1500),                                # c = 3000 < shard rows: large select per shard
    (30000, [0, 9000, 19000, 30000], 1025),                                # smallest k on the global-memory path
    (12000, [0, 5000, 12000], 6000),                                       # c == n_total: everything is a candidate
    (4000, [0, 2500, 4000], 4000),                                         # k == n_total
])
def test_sharded_large_k_equals_single_device_and_oracle(n, bounds, k):
    """candidates per shard + merge == ONE search over the whole corpus, bit for bit, at candidate counts beyond what the
    select / merge kernels sort in LDS; and the answer is the oracle's (near-tie rules: with hundreds of results
    some adjacent pair is always closer than the gap)."""
    import torch
    eng = _engine()
    d, eta, pref = 64, 0.3, 0.1
    raw = orc.synth_corpus(n, d, seed=n + k)
    cols = orc.synth_payload_columns(n, seed=n + k)
    Q = orc.synth_queries(3, d, seed=k)
    whole = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    ids_w, sc_w = whole.search(Q, k, eta, pref)
    c = min(2 * k, n)
    qd = torch.from_numpy(Q).cuda()
    lists = torch.stack([sh.candidates_device(qd, c) for sh in _shards(eng, raw, cols, bounds)])
    recs = eng.records_to_numpy(lists)
    for s, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        real = min(c, hi - lo)
        assert np.all(recs["id"][s][:, :real] >= lo) and np.all(recs["id"][s][:, :real] < hi)
        assert np.all(recs["id"][s][:, real:] == -1)                              # padding behind a short shard's rows
        assert np.all(np.diff(recs["sim"][s][:, :real], axis=1) <= 0)              # sorted by similarity
    ids, sc = eng.merge_rerank_device(lists, c, k, eta, pref)
    assert np.array_equal(ids.cpu().numpy(), ids_w) and np.array_equal(sc.cpu().numpy(), sc_w)
    E = whole.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for j in range(Q.shape[0]):
        _, msg = compare_query(E, Q[j], dewi32, ent32, k, eta, pref, "cosine", ids_w[j], sc_w[j])
        assert msg is None, (j, msg)
        ref_ids, _ = orc.search(E, Q[j], dewi32, ent32, k, eta, pref)
        assert np.mean(ids_w[j] == ref_ids) > 0.98                                 # near-tie swaps only
        if k == n:
            assert sorted(ids_w[j].tolist()) == list(range(n))


def test_scan_finish_split_with_large_candidate_counts():
    """dewi_knn_scan + dewi_knn_finish (PipelinedSearcher, two streams) beyond 2048 candidates: final results and shard
    records equal the one-call entry points."""
    import torch
    eng = _engine()
    n, d, k = 20_000, 128, 1300
    raw = orc.synth_corpus(n, d, seed=5)
    cols = orc.synth_payload_columns(n, seed=5)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], id_offset=70_000)
    Q = torch.from_numpy(orc.synth_queries(4, d, seed=6)).cuda()
    want_ids, want_sc = c.search_device(Q, k, 0.4, 0.0)
    pipe = eng.PipelinedSearcher(c, k, 0.4, 0.0, n_queries=2)
    ids = torch.empty((2, 2, k), dtype=torch.int64, device="cuda")
    sc = torch.empty((2, 2, k), dtype=torch.float32, device="cuda")
    for j in range(2):
        pipe.submit(Q[2 * j:2 * j + 2], ids[j], sc[j])
    pipe.drain()
    # dewi_knn_finish adds the shard's id offset, the one-call search returns local rows
    assert torch.equal(ids.view(4, k), want_ids + 70_000) and torch.equal(sc.view(4, k), want_sc)
    recs = torch.empty((2, 2, 2 * k, 4), dtype=torch.int32, device="cuda")
    pipe2 = eng.PipelinedSearcher(c, k, 0.4, 0.0, n_queries=2, n_candidates=2 * k)
    for j in range(2):
        pipe2.submit(Q[2 * j:2 * j + 2], out_records=recs[j])
    pipe2.drain()
    assert torch.equal(recs.view(4, 2 * k, 4), c.candidates_device(Q, 2 * k))
    assert int(recs[..., 3].min()) >= 70_000


def test_large_merge_carries_the_refusal_marker_and_rejects_a_short_workspace():
    """A shard whose batched path refused a query marks its records with id -2: the large merge hands the query back
    as unanswered (ids -1), like the LDS merge; and the C ABI checks the workspace it now needs."""
    import torch
    from dewi import _native as nat
    eng = _engine()
    n, d, k = 6000, 64, 600
    raw = orc.synth_corpus(n, d, seed=1)
    cols = orc.synth_payload_columns(n, seed=1)
    c = 2 * k
    qd = torch.from_numpy(orc.synth_queries(2, d, seed=2)).cuda()
    lists = torch.stack([sh.candidates_device(qd, c) for sh in _shards(eng, raw, cols, [0, 2000, 4000, 6000])])
    good_ids, _ = eng.merge_rerank_device(lists, c, k, 0.3, 0.0)
    marked = lists.clone()
    marked[1, 0, :, 3] = -2                                   # shard 1 refused query 0
    ids, sc = eng.merge_rerank_device(marked, c, k, 0.3, 0.0)
    assert torch.all(ids[0] == -1) and torch.all(torch.isnan(sc[0]))
    assert torch.equal(ids[1], good_ids[1])
    lib = nat.load_library()
    need = int(lib.dewi_merge_workspace_bytes(3, 2, c, c))
    assert need > 0 and int(lib.dewi_merge_workspace_bytes(3, 2, 20, 20)) == 0
    small = torch.empty(need - 8, dtype=torch.uint8, device="cuda")
    rc = lib.dewi_merge_rerank(nat.ptr(lists), 3, 2, c, c, k, 0.3, 0.0, nat.ptr(ids), nat.ptr(sc), nat.ptr(small), need - 8,
                               nat.stream_ptr())
    assert rc == nat.ERR_WORKSPACE and "workspace" in nat.last_error()


@pytest.mark.parametrize("dim,k,b,qscale", [(768, 10, 33, 1.0), (512, 50, 8, 1.0), (256, 128, 40, 1.0), (768, 10, 12, 20.0),
                                            (384, 10, 12, 1.0), (640, 50, 33, 1.0), (160, 10, 6, 1.0),       # round 4: partial last chunk
                                            (300, 10, 12, 1.0), (132, 10, 33, 1.0), (700, 50, 8, 1.0)])      # ... that ends inside a wave's slice
def test_l2_exact_refine_equals_the_one_query_search(dim, k, b, qscale):
    """l2 batches over an fp32 corpus run on the matrix cores in exact-refine mode: the pass's score has an absolute error
    bound, the candidate cut is widened by it and the candidates are re-scored with the row kernels' arithmetic — so a
    batch must equal the one-query searches BIT FOR BIT (ids and scores), whatever the norms (qscale 20: ||q||^2 ~ 300 000,
    where the unrefined matrix-core score is off by ~0.05), at every supported dim, and agree with the oracle."""
    import torch
    eng = _engine()
    n = 66_000 + dim
    rng = np.random.default_rng(dim + k)
    raw = orc.synth_corpus(n, dim, seed=dim + k) * rng.uniform(0.5, 4.0, size=(n, 1)).astype(np.float32)
    Q = (orc.synth_queries(b, dim, seed=k) * rng.uniform(0.5, 2.0, size=(b, 1)) * qscale).astype(np.float32)
    Q[0] = raw[123]                                                   # an exact duplicate
    cols = orc.synth_payload_columns(n, seed=dim)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space="l2")
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    ids, sc = c.search(Q, k, 0.3, 0.1)
    assert ids.min() >= 0
    for j in range(b):
        i1, s1 = c.search(Q[j], k, 0.3, 0.1)
        assert np.array_equal(ids[j], i1[0]) and np.array_equal(sc[j], s1[0]), j
    assert ids[0][0] == 123 or 123 in ids[0]
    check = list(range(0, b, max(1, b // 8)))
    for j in check:
        _, msg = compare_query(raw, Q[j], dewi32, ent32, k, 0.3, 0.1, "l2", ids[j], sc[j], exact_gaps=False)
        assert msg is None, (j, msg)
    # shard records through the same mode: candidates + merge == whole
    qd = torch.from_numpy(Q).cuda()
    cc = 2 * k
    recs = c.candidates_device(qd, cc)
    m_ids, m_sc = (t.cpu().numpy() for t in eng.merge_rerank_device(recs.unsqueeze(0), cc, k, 0.3, 0.1))
    # (at qscale 20 the error band holds more candidates than the sort: the pass refuses such a query and the repair inside
    # dewi_knn_candidates writes its records from the row kernels — nothing is marked, the merge answers every query)
    assert not c.refused_by_last_call().any() or qscale > 1.0
    assert m_ids.min() >= 0
    assert np.array_equal(m_ids, ids) and np.array_equal(m_sc, sc)


@pytest.mark.parametrize("dim,n,b,k", [(768, 70_001, 32, 10), (256, 131_073, 12, 10), (512, 80_000, 40, 50)])
def test_default_l2_batches_over_fp32_vs_oracle(dim, n, b, k):
    """The DEFAULT mode of space="l2" batches over an fp32 corpus (matrix-core pass + exact refinement; nothing opted in)
    through the oracle harness at the floors the unrefined tests had before round 3 (80 % decisive at k <= 10, 25 % above):
    rows and queries of very different lengths, as tests/test_hip_mfma_f32.py::test_mfma_l2_batched_vs_oracle."""
    import torch
    eng = _engine()
    rng = np.random.default_rng(dim + b)
    raw = orc.synth_corpus(n, dim, seed=dim + b) * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    Q = orc.synth_queries(b, dim, seed=b) * rng.uniform(0.5, 2.0, size=(b, 1)).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim + b)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space="l2")
    assert c.scan_kernel_name(b, k).startswith("mfma_scan_f32")          # the default really is the matrix-core pass
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, 0.3, 0.1)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0 and not np.isnan(sc).any()
    check_batch(raw, Q, dewi32, ent32, k, 0.3, 0.1, "l2", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.25,
                exact_gaps=False)


@pytest.mark.parametrize("bf16", [False, True])
def test_l2_batches_keep_near_duplicates_at_zero_distance(bf16):
    """The default l2 batch path must give what the reference's -sum((E - q)^2) gives (backends.py:434-436) where it
    matters most: queries that are (near-)duplicates of corpus rows with large norms.  On the matrix cores the score
    2<e,q> - ||e||^2 - ||q||^2 comes back as +-1e-4 noise at ||e||^2 ~ 500 — so over an fp32 corpus the pass runs in
    exact-refine mode (error-widened cut, candidates re-scored with the row kernels' arithmetic) and over a bf16 corpus
    the batch takes the exact row kernels: either way search_batch(Q)[j] == search(Q[j]) bit for bit, the duplicate's
    score is the reference's to the UNSCALED 1e-5 of north_star, and ranking among near neighbours follows the oracle."""
    import torch
    eng = _engine()
    n, d, b, k = 70_000, 256, 12, 10
    rng = np.random.default_rng(7)
    norms = rng.uniform(0.5, 30.0, size=(n, 1)).astype(np.float32)
    raw = orc.synth_corpus(n, d, seed=7) * norms                     # ||e|| from 0.5 to 30: ||e||^2 up to 900
    rows = rng.choice(n, size=b, replace=False)
    Q = raw[rows].copy()
    Q[b // 2:] += rng.normal(0, 1e-3, size=(b - b // 2, d)).astype(np.float32)   # second half: near, not exact, duplicates
    for j in range(b):                                                # a few more close rows around every query
        near = rng.choice(n, size=3, replace=False)
        raw[near] = Q[j] + rng.normal(0, 3e-3, size=(3, d)).astype(np.float32)
    raw[rows[: b // 2]] = Q[: b // 2]
    cols = orc.synth_payload_columns(n, seed=7)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space="l2")
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    if bf16:
        c = c.to_bf16()
        E = c.emb.float().cpu().numpy()
        Qo = np.stack([orc.bf16_round(q) for q in Q])                # l2: queries are rounded, not normalised
    else:
        E, Qo = raw, Q
    ids, sc = c.search(Q, k, 0.0, 0.0)                               # eta = 0: the score IS the similarity
    one = [c.search(Q[j], k, 0.0, 0.0) for j in range(b)]
    for j in range(b):
        assert np.array_equal(ids[j], one[j][0][0]) and np.array_equal(sc[j], one[j][1][0]), j   # batch == single, bit for bit
        ref_ids, ref_sc = orc.search_prepared(E, Qo[j], dewi32, ent32, k, 0.0, 0.0, "l2") if bf16 else \
            orc.search(E, Qo[j], dewi32, ent32, k, 0.0, 0.0, "l2")
        assert np.max(np.abs(sc[j].astype(np.float64) - ref_sc.astype(np.float64))[:4]) <= 1e-5, (j, sc[j][:4], ref_sc[:4])
        if not bf16:                                                 # (bf16 rounding makes the planted rows exact ties)
            assert ids[j][0] == ref_ids[0]
        if j < b // 2 and not bf16:
            assert ids[j][0] == rows[j] and sc[j][0] == 0.0           # an exact duplicate is at distance exactly 0
        assert set(ids[j][:4].tolist()) == set(ref_ids[:4].tolist())  # the duplicate and its three planted neighbours


def test_c5_device_resident_pipeline_equals_the_host_array_path(monkeypatch):
    """Config C5 through the Python layer on CUDA tensors (the reference's caller: pipelines.py:180-223): I_hat written
    into its row of the [7][N] signal table -> fit on the table's rows -> score -> fp32 dewi column -> column ingest of
    the device embedding block -> build -> search.  While that runs, every device-to-host copy of more than 4096 elements
    is an error; the answers must equal the host-array path's bit for bit (medians, MADs, scores, search results)."""
    import torch
    from dewi import signals
    from dewi.index import DewiIndex
    from dewi.scorer import DewiScorer
    from dewi.types import SIGNAL_FIELDS
    n, d, k = 1_000_000, 512, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    A = torch.randn((n, d), generator=g, device="cuda")
    g.manual_seed(43)
    B = torch.randn((n, d), generator=g, device="cuda")
    B[::5] = A[::5] * 0.6 + B[::5] * 0.4
    rs = np.random.RandomState(5)
    sig = np.stack([rs.gamma(2, 0.5, n), rs.gamma(2, 0.5, n) * 1.5, rs.gamma(2, 0.3, n), rs.gamma(2, 0.3, n) * 1.5,
                    np.zeros(n), rs.beta(1, 5, n), rs.beta(1, 10, n)]).astype(np.float32)
    A_host, B_host = A.cpu().numpy(), B.cpu().numpy()           # taken BEFORE the device build normalises A in place
    Q = np.random.RandomState(6).randn(4, d).astype(np.float32)
    ids = [f"doc_{i:07d}" for i in range(n)]
    S = torch.from_numpy(sig).cuda()                             # the signal table, [7][N] on the device

    # ---- device-resident flow, with large device-to-host copies forbidden
    real_cpu, real_tolist = torch.Tensor.cpu, torch.Tensor.tolist

    def guarded(real):
        def f(self, *a, **kw):
            assert not (self.is_cuda and self.numel() > 4096), f"a {tuple(self.shape)} tensor was copied to the host"
            return real(self, *a, **kw)
        return f
    monkeypatch.setattr(torch.Tensor, "cpu", guarded(real_cpu))
    monkeypatch.setattr(torch.Tensor, "tolist", guarded(real_tolist))
    signals.cross_modal_similarity(A, B, return_device=True, out=S[4])
    cols = {name: S[j] for j, name in enumerate(SIGNAL_FIELDS)}
    scorer = DewiScorer()
    scorer.fit_stats_columns(cols)                               # in place on the table's rows: two kernels, no sync
    f64_dev, dewi32_dev = scorer.score_batch_device(cols)        # statistics read by the kernel from the device
    index = DewiIndex(dim=d, use_ann=False, rerank_eta=0.3)
    index.add_batch_columns(ids, A, {"dewi": dewi32_dev, "ht_mean": S[0], "ht_q90": S[1], "hi_mean": S[2], "hi_q90": S[3],
                                     "I_hat": S[4], "redundancy": S[5], "noise": S[6]})
    index.build()
    assert index._backend._corpus.emb.data_ptr() == A.data_ptr()            # the caller's block, used in place
    res_dev = index.search_batch(Q, k=k)
    monkeypatch.undo()

    # ---- the same through host arrays
    ihat_h = signals.cross_modal_similarity(A_host, B_host)
    assert np.array_equal(ihat_h, S[4].cpu().numpy())
    sig[4] = ihat_h
    cols_h = {name: sig[j] for j, name in enumerate(SIGNAL_FIELDS)}
    scorer_h = DewiScorer()
    scorer_h.fit_stats_columns(cols_h)
    assert scorer.stats.medians == scorer_h.stats.medians and scorer.stats.mads == scorer_h.stats.mads
    dewi_h = scorer_h.score_batch(cols_h)
    assert np.array_equal(dewi_h, f64_dev.cpu().numpy())
    assert np.array_equal(dewi_h.astype(np.float32), dewi32_dev.cpu().numpy())
    med, mad = orc.robust_fit(cols_h)                            # and the oracle: statistics bit-exact, scores to 2 ulp
    assert scorer.stats.medians == med and scorer.stats.mads == mad
    ref = orc.score({key: v.astype(np.float64) for key, v in cols_h.items()}, med, mad)
    assert np.max(np.abs(dewi_h - ref) / ref) < 1e-15
    index_h = DewiIndex(dim=d, use_ann=False, rerank_eta=0.3)
    index_h.add_batch_columns(ids, A_host, {"dewi": dewi_h, "ht_mean": sig[0], "ht_q90": sig[1], "hi_mean": sig[2],
                                            "hi_q90": sig[3], "I_hat": sig[4], "redundancy": sig[5], "noise": sig[6]})
    res_h = index_h.search_batch(Q, k=k)
    for rd, rh in zip(res_dev, res_h):
        assert [(a[0], a[1]) for a in rd] == [(a[0], a[1]) for a in rh]
        for a, b in zip(rd, rh):                                  # payloads made from the device columns: fp32 values
            assert a[2].dewi == float(np.float32(b[2].dewi)) and a[2].ht_mean == b[2].ht_mean and a[2].I_hat == b[2].I_hat
    again = index.search_batch(Q[:1], k=k)[0]
    assert all(x[2] is y[2] for x, y in zip(again, res_dev[0]))  # the same Payload objects from then on


def test_redundancy_top1_at_full_size():
    """Config C5's redundancy leg at its full size (1 M x 512 text / image embeddings on the device): a self-join through
    the batched bf16 matrix-core path.  The reduction is this build's own definition (the reference defines none:
    SURVEY §8 F3, parity unpinned), so it is checked by properties and against a brute-force product on 64 sampled rows."""
    import torch
    from dewi import signals
    n, d = 1_000_000, 512
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    T = torch.randn((n, d), generator=g, device="cuda")
    I = torch.randn((n, d), generator=g, device="cuda")
    I[1000:1064] = T[5000:5064] * 0.9 + I[1000:1064] * 0.1       # texts 5000.. have a close image of ANOTHER document
    r = signals.redundancy_top1(T, I, return_device=True)
    assert r.shape == (n,) and r.is_cuda and bool(torch.isfinite(r).all())
    assert float(r.min()) > 0.1 and float(r.max()) <= 1.0 + 1e-3  # the best of a million random directions in 512 dims
    assert float(r[5000:5064].min()) > 0.9
    In = torch.nn.functional.normalize(I, dim=1).bfloat16().float()
    sample = [0, 1, 17, 5000, 5063, 999_999] + np.random.RandomState(3).randint(0, n, 58).tolist()
    for i in sample:
        q = torch.nn.functional.normalize(T[i], dim=0).bfloat16().float()
        sims = In @ q
        sims[i] = -2.0                                            # any OTHER document
        assert abs(float(sims.max()) - float(r[i])) < 2e-3, i


def test_finish_repairs_a_refused_query_with_the_shards_id_offset():
    """dewi_knn_finish answers a query the matrix-core pass refused on the exact kernels itself (ABI 5): the repaired row
    carries the shard's id offset like every other row it writes — nothing is left for ``drain`` to do."""
    import torch
    eng = _engine()
    n, dim, k, b, off = 100_000, 256, 10, 40, 5_000_000
    raw = orc.synth_corpus(n, dim, seed=3)
    raw[50_000:90_000] = raw[7]
    cols = orc.synth_payload_columns(n, seed=3)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], id_offset=off).to_bf16()
    Q = orc.synth_queries(b, dim, seed=4)
    Q[5] = raw[7]                                                 # overflows its survivor segments
    want_ids, want_sc = c.search(Q, k, 0.3, 0.1)                  # global ids
    assert want_ids.min() >= off
    assert c.refused_by_last_call().tolist() == [j == 5 for j in range(b)]
    q_dev = torch.from_numpy(Q).cuda()
    one_ids, one_sc = c.search(Q[5], k, 0.3, 0.1)
    assert np.array_equal(want_ids[5], one_ids[0]) and np.array_equal(want_sc[5], one_sc[0])
    pipe = eng.PipelinedSearcher(c, k, 0.3, 0.1, n_queries=b)
    ids = torch.empty((b, k), dtype=torch.int64, device="cuda")
    sc = torch.empty((b, k), dtype=torch.float32, device="cuda")
    pipe.submit(q_dev, ids, sc)
    pipe.finish_stream.synchronize()                              # no drain-side repair exists any more
    assert np.array_equal(ids.cpu().numpy(), want_ids) and np.array_equal(sc.cpu().numpy(), want_sc)


def test_scorer_row_api_when_the_callers_list_grew_after_fit():
    """fit_stats(rows) keeps the caller's list to serve score(sig) from one launch; dicts APPENDED to that list afterwards
    are not in the fitted table and must simply be scored (the reference scores any dict), not raise."""
    from dewi.scorer import DewiScorer
    cols = orc.synth_payload_columns(300, seed=8)
    rows = [{key: float(np.float32(cols[key][i])) for key in orc.SIGNAL_KEYS} for i in range(300)]
    fitted = rows[:200]
    s = DewiScorer()
    s.fit_stats(fitted)
    first = [s.score(r) for r in fitted[:5]]
    fitted.extend(rows[200:])                                     # the list the scorer holds grows
    later = [s.score(r) for r in fitted]                          # sequential scan walks past the fitted length
    assert later[:5] == first
    med, mad = s.stats.medians, s.stats.mads
    ref = orc.score({key: np.array([r[key] for r in fitted], np.float64) for key in orc.SIGNAL_KEYS}, med, mad)
    assert np.max(np.abs(np.array(later) - ref) / ref) < 1e-15


def test_device_ingest_paths_equal_host_ingest(tmp_path):
    """ExactIndex.add_batch_columns with CUDA tensors on its other paths — several device blocks, a host block in
    between, rows appended to a built index, copy=True, space l2 (no normalisation), save / load and iteration over
    the payload store afterwards — each equal to the same rows ingested as host arrays."""
    import torch
    from dewi.backends import ExactIndex
    n, d, k = 3000, 64, 7
    raw = orc.synth_corpus(n, d, seed=21)
    cols = orc.synth_payload_columns(n, seed=21)
    ids = [f"d{i}" for i in range(n)]
    Q = orc.synth_queries(5, d, seed=22)
    fields = ("dewi", "ht_mean", "hi_mean", "noise")
    host_cols = {f: cols[f] for f in fields}

    def dev_cols(lo, hi):
        return {f: torch.from_numpy(cols[f][lo:hi]).cuda() for f in fields}          # float64 device columns

    for space in ("cosine", "l2"):
        ref = ExactIndex(dim=d, space=space)
        ref.add_batch_columns(ids, raw, host_cols)
        want = ref.search_batch(Q, k, 0.3, 0.2)
        # (a) three blocks: device, host, device
        a = ExactIndex(dim=d, space=space)
        a.add_batch_columns(ids[:1000], torch.from_numpy(raw[:1000]).cuda(), dev_cols(0, 1000))
        a.add_batch_columns(ids[1000:1800], raw[1000:1800], {f: cols[f][1000:1800] for f in fields})
        a.add_batch_columns(ids[1800:], torch.from_numpy(raw[1800:]).cuda(), dev_cols(1800, n))
        got = a.search_batch(Q, k, 0.3, 0.2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), space
        # (a') ONE block whose payload columns are mixed — some CUDA tensors, some host arrays (round 3's advice: the
        # device-covered build used to call .to() on an ndarray)
        m = ExactIndex(dim=d, space=space)
        mixed = dev_cols(0, n)
        mixed["hi_mean"] = cols["hi_mean"]                       # a host column among device ones
        mixed["noise"] = cols["noise"]
        m.add_batch_columns(ids, torch.from_numpy(raw).cuda(), mixed, copy=True)
        got = m.search_batch(Q, k, 0.3, 0.2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), space
        # (b) rows appended to a built index (stored rows are not normalised twice)
        b = ExactIndex(dim=d, space=space)
        b.add_batch_columns(ids[:2000], torch.from_numpy(raw[:2000]).cuda(), dev_cols(0, 2000))
        b.build()
        assert b.search_batch(Q, k, 0.3, 0.2)[0].max() < 2000
        b.add_batch_columns(ids[2000:], torch.from_numpy(raw[2000:]).cuda(), dev_cols(2000, n))
        got = b.search_batch(Q, k, 0.3, 0.2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), space
        # (c) copy=True leaves the caller's tensor alone; the default normalises it in place (cosine)
        mine = torch.from_numpy(raw).cuda()
        c = ExactIndex(dim=d, space=space)
        c.add_batch_columns(ids, mine, dev_cols(0, n), copy=True)
        c.build()
        assert torch.equal(mine.cpu(), torch.from_numpy(raw))
        got = c.search_batch(Q, k, 0.3, 0.2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), space
        # results carry Payload objects built from the device columns; whole-store views bring the block over
        res = c.results_for(got[0][:1], got[1][:1])[0]
        assert [r[0] for r in res] == [ids[i] for i in got[0][0]]
        assert all(r[2].noise == cols["noise"][i] and r[2].dewi == cols["dewi"][i] for r, i in zip(res, got[0][0]))
        assert len(c._payloads) == n and sum(1 for _ in c._payloads.values()) == n
        # (d) save / load round trip of a device-ingested index, then refresh_payloads after an in-place edit
        c.save(tmp_path / space)
        back = ExactIndex.load(tmp_path / space)
        got = back.search_batch(Q, k, 0.3, 0.2)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), space
        top = int(want[0][0][0])
        c._payloads[ids[top]].dewi = -5.0
        c.refresh_payloads()
        assert c.search_batch(Q[:1], k, 0.9, 0.0)[0][0][0] != top or k == 1     # a very low dewi at eta 0.9 drops the row


def test_bf16_dim768_shards_are_bit_equal_on_even_boundaries():
    """Found by scripts/fuzz_parity.py: the one-query kernel of a bf16 corpus scans 1536-byte rows (dim 768) in PAIRS and
    sums the first and the second row of a pair in different lane orders, so a shard that starts on an ODD row scores its
    rows with the other order: equal to the whole-corpus search to summation noise, not bit for bit.  shard_bounds cuts on
    even rows, where candidates + merge == whole holds bit for bit; an odd cut is still the oracle's answer."""
    import torch
    from dewi.sharded import shard_bounds
    eng = _engine()
    n, d, k, eta, pref = 7778, 768, 40, 0.0, 0.0
    raw = orc.synth_corpus(n, d, seed=91)
    cols = orc.synth_payload_columns(n, seed=91)
    Q = orc.synth_queries(1, d, seed=92)
    whole = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"]).to_bf16()
    qd = torch.from_numpy(Q).cuda()
    w_ids, w_sc = whole.search_device(qd, k, eta, pref)
    c = 2 * k

    def sharded(cuts):
        lists = [eng.DeviceCorpus(whole.emb[lo:hi], whole.dewi32[lo:hi], whole.ent32[lo:hi], "cosine", id_offset=lo)
                 .candidates_device(qd, c) for lo, hi in zip(cuts[:-1], cuts[1:])]
        return eng.merge_rerank_device(torch.stack(lists), c, k, eta, pref)

    for world in (2, 3, 5, 8):
        b = shard_bounds(n, world)
        ids, sc = sharded([lo for lo, _ in b] + [n])
        assert torch.equal(ids, w_ids) and torch.equal(sc, w_sc), world
    ids, sc = sharded([0, 1637, 6445, 6861, n])                    # odd cuts: the fuzz case
    E = whole.emb.float().cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    from parity import device_prepared_queries
    Qp = device_prepared_queries(Q)
    _, msg = compare_query(E, Qp[0], dewi32, ent32, k, eta, pref, "cosine", ids[0].cpu().numpy(), sc[0].cpu().numpy(),
                           gap=1e-6, score_tol=1e-5, prepared=True)
    assert msg is None, msg
    assert torch.allclose(torch.sort(sc, dim=1).values, torch.sort(w_sc, dim=1).values, rtol=0, atol=1e-6)


@pytest.mark.parametrize("dim,n,b,k", [(768, 70_000, 256, 10), (768, 100_003, 300, 100), (256, 131_072, 64, 50), (512, 66_000, 33, 1),
                                       (768, 80_000, 32, 10), (512, 70_000, 2, 5), (256, 66_000, 8, 128), (768, 66_001, 5, 100),
                                       (768, 120_526, 65, 129),    # (the pre-selection's own workspace size, a fuzz find)
                                       (1024, 70_000, 40, 10), (1024, 66_000, 8, 50),    # dim 1024 / 1536: depth-split pass in groups of 32
                                       (1536, 66_000, 33, 10), (1536, 70_001, 4, 100),
                                       # round 4: widths outside the dim = 256 U set (the re-scoring repeats scan_rows_any's arithmetic):
                                       # 384 / 640 with more than 32 queries = the 256-query pass; otherwise the depth-split pass with a
                                       # partial last chunk over the shadow
                                       (384, 70_000, 256, 10), (384, 66_000, 12, 50), (640, 66_001, 40, 10), (160, 80_000, 33, 10),
                                       (992, 66_000, 8, 10), (1280, 65_600, 5, 20),
                                       # ... dim % 8 == 0 from 136 columns: rows that end inside a wave's 32-column slice
                                       (200, 70_000, 12, 10), (1000, 65_600, 33, 10), (136, 66_000, 5, 10), (264, 66_000, 40, 20)])
def test_fp32_corpus_with_bf16_shadow_equals_the_one_query_search(dim, n, b, k):
    """fp32 corpus + bf16 shadow (dewi_knn_rerank_f32_shadow): a batch of cosine queries runs a matrix-core pass over the
    shadow as a pre-selection (2-32 queries: the depth-split pass in its bf16 geometry; more: the 256-query pass) and
    re-scores the candidates from the fp32 rows with the row kernels' arithmetic — so it must equal the one-query fp32
    searches BIT FOR BIT (ids and scores) and the oracle; the plain fp32 batch path (32 queries per pass, three-piece cut;
    row kernels below 5 queries) agrees up to its own summation order."""
    import torch
    eng = _engine()
    raw = orc.synth_corpus(n, dim, seed=n % 1000 + dim)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = orc.synth_queries(b, dim, seed=b + k)
    Q[min(3, b - 1)] = raw[77] * 3.0                                # a scaled copy of a row: cosine 1
    plain = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    c = eng.DeviceCorpus(plain.emb, plain.dewi32, plain.ent32, "cosine").enable_bf16_shadow()
    assert c.shadow is not None and c.shadow.dtype == torch.bfloat16
    ids, sc = c.search(Q, k, 0.3, 0.1)
    assert ids.min() >= 0
    E = plain.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for j in sorted(set(j for j in (0, 3, 31, 32, b // 2, b - 1) if j < b)):
        i1, s1 = plain.search(Q[j], k, 0.3, 0.1)                     # one query: the row kernels
        assert np.array_equal(ids[j], i1[0]) and np.array_equal(sc[j], s1[0]), j
        _, msg = compare_query(E, Q[j], dewi32, ent32, k, 0.3, 0.1, "cosine", ids[j], sc[j], exact_gaps=False)
        assert msg is None, (j, msg)
    ids_p, sc_p = plain.search(Q, k, 0.3, 0.1)                       # the unshadowed batch path
    assert np.mean(ids_p == ids) > 0.98 and np.allclose(np.sort(sc_p, axis=1), np.sort(sc, axis=1), rtol=0, atol=2e-6)
    ids2, sc2 = c.search(Q, k, 0.3, 0.1)                             # deterministic
    assert np.array_equal(ids2, ids) and np.array_equal(sc2, sc)


@pytest.mark.parametrize("dim,n,k", [(768, 70_000, 10), (256, 131_073, 128), (512, 66_000, 1), (768, 80_001, 16), (256, 70_000, 17),
                                     (1024, 70_000, 10), (1024, 66_000, 64), (1536, 70_000, 10),
                                     (384, 70_000, 10), (640, 66_000, 40),           # round 4: the depth-split pass, partial last chunk
                                     (200, 70_000, 10), (1000, 66_000, 40)])         # ... dim % 8 == 0, not % 32
def test_one_query_through_the_bf16_shadow_equals_the_fp32_row_scan(dim, n, k):
    """enable_bf16_shadow(single_query=True): ONE query runs a pass over the shadow — cuts of up to 32 rows (k <= 16) the bf16 row
    kernel with per-workgroup lists long enough for the error band, larger cuts the depth-split pass — + the exact re-scoring;
    ids and scores equal the plain fp32 one-query search bit for bit, for many queries (incl. an exact copy of a row),
    through the blocking API and on device tensors; without the switch one query keeps the fp32 row scan."""
    import torch
    eng = _engine()
    raw = orc.synth_corpus(n, dim, seed=n % 1000 + dim + 1)
    cols = orc.synth_payload_columns(n, seed=dim + 1)
    Q = orc.synth_queries(24, dim, seed=k + 5)
    Q[5] = raw[4242] * 0.5
    plain = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    c = eng.DeviceCorpus(plain.emb, plain.dewi32, plain.ent32, "cosine").enable_bf16_shadow(single_query=True)
    assert c.shadow_min_batch == 1
    E = plain.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for j in range(24):
        ids, sc = c.search(Q[j], k, 0.3, 0.1)
        i1, s1 = plain.search(Q[j], k, 0.3, 0.1)
        assert ids.shape == (1, k) and ids.min() >= 0
        assert np.array_equal(ids, i1) and np.array_equal(sc, s1), j
        if j < 6:
            _, msg = compare_query(E, Q[j], dewi32, ent32, k, 0.3, 0.1, "cosine", ids[0], sc[0], exact_gaps=False)
            assert msg is None, (j, msg)
    qd = torch.from_numpy(Q).cuda()
    d_ids, d_sc = c.search_device(qd[5:6].contiguous(), k, 0.3, 0.1)
    p_ids, p_sc = c.search_device(qd[5:6].contiguous(), k, 0.3, 0.1, use_shadow=False)
    assert torch.equal(d_ids, p_ids) and torch.equal(d_sc, p_sc)
    assert int(d_ids[0, 0]) == 4242 or k > 1                          # eta = 0.3: the copy leads unless DEWI outweighs it
    two = eng.DeviceCorpus(plain.emb, plain.dewi32, plain.ent32, "cosine").enable_bf16_shadow()
    assert two.shadow_min_batch == 2
    # the one-query call really reads the shadow: with the shadow zeroed every row ties at 0, the survivors overflow, the
    # pass refuses the query — and the repair launches of the same call answer it from the fp32 rows (ABI 5) — while the
    # corpus without the switch never looks at the shadow
    c.shadow.zero_()
    z_ids, z_sc = c.search_device(qd[5:6].contiguous(), k, 0.3, 0.1)
    assert c.refused_by_last_call().tolist() == [True]
    assert torch.equal(z_ids, p_ids) and torch.equal(z_sc, p_sc)
    two.shadow.zero_()
    t_ids, t_sc = two.search_device(qd[5:6].contiguous(), k, 0.3, 0.1)
    assert torch.equal(t_ids, p_ids) and torch.equal(t_sc, p_sc)
    ids, sc = c.search(Q[5], k, 0.3, 0.1)
    assert np.array_equal(ids, p_ids.cpu().numpy()) and np.array_equal(sc, p_sc.cpu().numpy())


def test_exact_index_with_batch_shadow_answers_as_the_plain_index():
    """ExactIndex(batch_shadow=True[, shadow_single_query=True]) — the reference's API (backends.py:369-481: add / build /
    search) with the additive switches: search() and search_batch() return what the plain index returns, id for id and bit
    for bit, whether the rows arrive as host arrays or as a device block."""
    import torch
    from dewi.index import ExactIndex
    n, dim, k = 70_000, 256, 10
    raw = orc.synth_corpus(n, dim, seed=91)
    cols = orc.synth_payload_columns(n, seed=92)
    ids = [f"d{i}" for i in range(n)]
    Q = orc.synth_queries(40, dim, seed=93)
    plain = ExactIndex(dim=dim, space="cosine")
    plain.add_batch_columns(ids, raw, cols)
    plain.build()
    # one query at a time: the row kernels, whose arithmetic the shadow path's re-scoring repeats (the plain 40-query batch takes
    # the fp32 matrix-core pass: same ids, another summation order)
    one = [plain.search_batch(Q[j:j + 1], k=k, eta=0.3, entropy_pref=0.1) for j in range(Q.shape[0])]
    want_rows, want_sc = np.concatenate([o[0] for o in one]), np.concatenate([o[1] for o in one])
    b_rows, b_sc = plain.search_batch(Q, k=k, eta=0.3, entropy_pref=0.1)
    assert np.mean(b_rows == want_rows) > 0.98 and np.allclose(np.sort(b_sc, axis=1), np.sort(want_sc, axis=1), rtol=0, atol=2e-6)
    for single, on_device in ((False, False), (True, False), (True, True)):
        idx = ExactIndex(dim=dim, space="cosine", batch_shadow=True, shadow_single_query=single)
        if on_device:
            idx.add_batch_columns(ids, torch.from_numpy(raw).cuda(), {kk: torch.from_numpy(np.asarray(v)).cuda() for kk, v in cols.items()})
        else:
            idx.add_batch_columns(ids, raw, cols)
        idx.build()
        assert idx._corpus.shadow is not None and idx._corpus.shadow_min_batch == (1 if single else 2)
        rows, sc = idx.search_batch(Q, k=k, eta=0.3, entropy_pref=0.1)
        assert np.array_equal(rows, want_rows) and np.array_equal(sc, want_sc)
        for j in (0, 7):
            got = idx.search(Q[j], k=k, eta=0.3, entropy_pref=0.1)
            ref = plain.search(Q[j], k=k, eta=0.3, entropy_pref=0.1)
            assert [r[0] for r in got] == [r[0] for r in ref] and [r[1] for r in got] == [r[1] for r in ref]
            assert [r[0] for r in got] == [ids[r] for r in want_rows[j]]
    from dewi.index import DewiIndex
    fac = DewiIndex(dim=dim, space="cosine", use_ann=False, rerank_eta=0.3, entropy_pref=0.1, batch_shadow=True, shadow_single_query=True)
    fac.add_batch_columns(ids, raw, cols)
    fac.build()
    assert fac._backend._corpus.shadow is not None and fac._backend._corpus.shadow_min_batch == 1
    got = fac.search(Q[3], k=k)
    ref = plain.search(Q[3], k=k, eta=0.3, entropy_pref=0.1)
    assert [r[0] for r in got] == [r[0] for r in ref] and [r[1] for r in got] == [r[1] for r in ref]
    l2 = ExactIndex(dim=dim, space="l2", batch_shadow=True)          # the switch is a no-op outside cosine
    l2.add_batch_columns(ids[:1000], raw[:1000], {kk: np.asarray(v)[:1000] for kk, v in cols.items()})
    l2.build()
    assert l2._corpus.shadow is None


def test_one_query_through_the_shadow_is_repaired_when_the_pass_refuses_it():
    """A corpus with 20 000 copies of one row: the pre-selection's survivors of a query that points at them overflow the
    segment, the pass refuses the query, and the repair launches behind it (same library call, same stream) answer it on
    the plain fp32 scan — the raw ABI call returns the same answer as the unshadowed corpus gives."""
    import torch
    eng = _engine()
    n, dim, k = 90_000, 256, 10
    raw = orc.synth_corpus(n, dim, seed=77)
    raw[10_000:30_000] = raw[5]
    cols = orc.synth_payload_columns(n, seed=78)
    plain = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    c = eng.DeviceCorpus(plain.emb, plain.dewi32, plain.ent32, "cosine").enable_bf16_shadow(single_query=True)
    q = raw[5] * 2.0
    raw_ids, raw_sc = c.search_device(torch.from_numpy(q[None]).cuda(), k, 0.3, 0.1)
    assert c.refused_by_last_call().tolist() == [True]            # 20 000 rows inside the error band: refused, then repaired
    i1, s1 = plain.search(q, k, 0.3, 0.1)
    assert np.array_equal(raw_ids.cpu().numpy(), i1) and np.array_equal(raw_sc.cpu().numpy(), s1)
    ids, sc = c.search(q, k, 0.3, 0.1)
    assert ids.min() >= 0 and np.array_equal(ids, i1) and np.array_equal(sc, s1)


def test_bf16_shadow_refuses_rows_that_are_not_normalised():
    import torch
    eng = _engine()
    emb = torch.randn((1000, 256), device="cuda") * 3.0
    z = torch.zeros(1000, device="cuda")
    with pytest.raises(ValueError, match="not normalised"):
        eng.DeviceCorpus(emb, z, z, "cosine").enable_bf16_shadow()
    with pytest.raises(ValueError, match="cosine"):
        eng.DeviceCorpus(emb, z, z, "l2").enable_bf16_shadow()


@pytest.mark.parametrize("switch", [{"DEWI_STAGE_KEYS": "512"}, {"DEWI_SELECT_THREADS": "256"}],
                         ids=["staging-512-keys", "more-segments-than-threads"])
def test_refine_routes_when_the_survivors_do_not_fit_the_staging(switch):
    """The exact-refine select stages a query's survivors in LDS; when there are more than it holds it takes the same
    steps over the segments in global memory instead of refusing.  A child process with the staging shrunk to 512 keys
    (DEWI_STAGE_KEYS, a test switch) runs the bf16-shadow batch and the l2 batch that way: both must still equal the
    one-query searches bit for bit.  Second switch (DEWI_SELECT_THREADS=256): the 256-query pass's 1024 segments then
    outnumber the select's threads — that route must refine too (it used to sort the approximate keys as they stood:
    round 3's advice), never return bf16 scores as the exact answer."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    code = r"""
import sys, numpy as np
sys.path.insert(0, r'%s'); sys.path.insert(0, r'%s')
import dewi_oracle as orc
from dewi import _engine as eng
n, d, b, k = 70_000, 256, 40, 20
raw = orc.synth_corpus(n, d, seed=5); cols = orc.synth_payload_columns(n, seed=5); Q = orc.synth_queries(b, d, seed=6)
plain = eng.DeviceCorpus.from_host(raw, cols['dewi'], cols['ht_mean'], cols['hi_mean'])
sh = eng.DeviceCorpus(plain.emb, plain.dewi32, plain.ent32, 'cosine').enable_bf16_shadow()
ids, sc = sh.search(Q, k, 0.3, 0.1)
for j in range(b):
    i1, s1 = plain.search(Q[j], k, 0.3, 0.1)
    assert np.array_equal(ids[j], i1[0]) and np.array_equal(sc[j], s1[0]), ('shadow', j)
l2 = eng.DeviceCorpus.from_host(raw * 2.5, cols['dewi'], cols['ht_mean'], cols['hi_mean'], space='l2')
ids, sc = l2.search(Q, k, 0.3, 0.1)
for j in range(b):
    i1, s1 = l2.search(Q[j], k, 0.3, 0.1)
    assert np.array_equal(ids[j], i1[0]) and np.array_equal(sc[j], s1[0]), ('l2', j)
print('OK')
""" % (repo / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd", repo / "oracle")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **switch))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stderr[-3000:]
