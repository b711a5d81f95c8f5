"""Full-size checks at BASELINE.json configs[1] (1M x 768 fp32, k=10, eta=0.3) and configs[2]
(bf16, 256 queries, k=100): properties that do not need the oracle on every query — sortedness,
determinism, shard-independence (3 ragged shards + merge == whole) — plus the oracle itself on a
handful of queries.
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import compare_query, device_prepared_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def corpus_1m():
    import torch
    from dewi import _engine as eng
    from dewi import _native as nat
    n, d = 1_000_000, 768
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    emb = torch.randn((n, d), generator=g, device="cuda")
    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(emb), nat.ptr(emb), n, d, nat.stream_ptr()))
    cols = orc.synth_payload_columns(n, seed=42)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    c = eng.DeviceCorpus(emb, torch.from_numpy(dewi32).cuda(), torch.from_numpy(ent32).cuda(), "cosine")
    Q = torch.randn((64, d), generator=g, device="cuda")
    return c, dewi32, ent32, Q


def test_c2_properties_and_oracle_sample(corpus_1m):
    import torch
    from dewi import _engine as eng
    c, dewi32, ent32, Q = corpus_1m
    k, eta = 10, 0.3
    ids = torch.stack([c.search_device(Q[j:j + 1], k, eta, 0.0)[0][0] for j in range(64)])
    sc = torch.stack([c.search_device(Q[j:j + 1], k, eta, 0.0)[1][0] for j in range(64)])
    assert torch.all(sc[:, :-1] >= sc[:, 1:])                                   # sorted
    assert torch.all((ids >= 0) & (ids < c.n_rows))
    assert all(len(set(r.tolist())) == k for r in ids)                          # no duplicates
    ids_4 = torch.cat([c.search_device(Q[j:j + 4], k, eta, 0.0)[0] for j in range(0, 64, 4)])   # NQ=4 passes == NQ=1 passes
    sc_4 = torch.cat([c.search_device(Q[j:j + 4], k, eta, 0.0)[1] for j in range(0, 64, 4)])
    assert torch.equal(ids_4, ids) and torch.equal(sc_4, sc)
    # all 64 at once: the fp32 matrix-core pass (two passes of 32 queries), whose fp32 sums run in another order:
    # same rows (no near-tie among these queries' top 11), scores to fp32 summation noise
    ids_b, sc_b = c.search_device(Q, k, eta, 0.0)
    assert ids_b.min().item() >= 0
    assert (ids_b == ids).float().mean().item() > 0.99 and torch.allclose(sc_b, sc, rtol=0, atol=2e-6)
    ids_again, sc_again = c.search_device(Q, k, eta, 0.0)                       # deterministic
    assert torch.equal(ids_again, ids_b) and torch.equal(sc_again, sc_b)
    E_host = c.emb.cpu().numpy()
    Qb_h, ib_h, sb_h = Q.cpu().numpy(), ids_b.cpu().numpy(), sc_b.cpu().numpy()
    for j in range(32, 64):                                                      # the matrix-core pass vs the oracle (a whole pass of 32)
        decisive, msg = compare_query(E_host, Qb_h[j], dewi32, ent32, k, eta, 0.0, "cosine", ib_h[j], sb_h[j], exact_gaps=False)
        assert msg is None, (j, msg)
    # shard-independence: 3 ragged shards (views of the same matrix) + merge == whole
    bounds = [0, 333_333, 700_001, c.n_rows]
    lists = []
    for s in range(3):
        lo, hi = bounds[s], bounds[s + 1]
        sh = eng.DeviceCorpus(c.emb[lo:hi], c.dewi32[lo:hi], c.ent32[lo:hi], "cosine", id_offset=lo)
        lists.append(sh.candidates_device(Q, 2 * k))
    m_ids, m_sc = eng.merge_rerank_device(torch.stack(lists), 2 * k, k, eta, 0.0)
    assert torch.equal(m_ids, ids_b) and torch.equal(m_sc, sc_b)      # same per-row sums whichever shard holds the row
    # the oracle on half of the queries (round 4: 32 instead of 8)
    E = E_host
    Qh, ih, sh_ = Q.cpu().numpy(), ids.cpu().numpy(), sc.cpu().numpy()
    for j in range(32):
        decisive, msg = compare_query(E, Qh[j], dewi32, ent32, k, eta, 0.0, "cosine", ih[j], sh_[j], exact_gaps=False)
        assert msg is None, (j, msg)


def test_c3_bf16_batched_properties(corpus_1m):
    import torch
    c, dewi32, ent32, Q = corpus_1m
    cb = c.to_bf16()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    Qb = torch.randn((256, c.dim), generator=g, device="cuda")
    k, eta = 100, 0.3
    ids, sc = cb.search_device(Qb, k, eta, 0.0)                                 # matrix-core path
    assert ids.min().item() >= 0 and not torch.isnan(sc).any()                  # no overflowed query
    assert torch.all(sc[:, :-1] >= sc[:, 1:])
    assert all(len(set(r.tolist())) == k for r in ids[:16])
    ids2, sc2 = cb.search_device(Qb, k, eta, 0.0)                               # deterministic despite the
    assert torch.equal(ids2, ids) and torch.equal(sc2, sc)                      # order survivors are stored in
    # agreement with the exact small-batch bf16 kernels (matrix-core path switched off; different summation
    # order)
    from dewi import _engine as eng
    for q0 in (0, 4):
        eng.tuning(0, 0, -1, 0)
        try:
            ids_s, sc_s = cb.search_device(Qb[q0:q0 + 4].contiguous(), k, eta, 0.0)
            torch.cuda.synchronize()
        finally:
            eng.tuning(0, 0, -1, 1)
        assert (ids_s == ids[q0:q0 + 4]).float().mean().item() > 0.97
        assert torch.allclose(torch.sort(sc_s, dim=1).values, torch.sort(sc[q0:q0 + 4], dim=1).values, atol=2e-5)
    # ... a batch of 40 takes the same 256-query kernel: same answers as inside the batch of 256, bit for bit;
    # a batch of 8 takes the depth-split pass (other summation order): same rows, scores to fp32 summation noise
    ids_40, sc_40 = cb.search_device(Qb[:40].contiguous(), k, eta, 0.0)
    assert torch.equal(ids_40, ids[:40]) and torch.equal(sc_40, sc[:40])
    ids_8, sc_8 = cb.search_device(Qb[:8].contiguous(), k, eta, 0.0)
    assert ids_8.min().item() >= 0
    assert (ids_8 == ids[:8]).float().mean().item() > 0.97
    assert torch.allclose(torch.sort(sc_8, dim=1).values, torch.sort(sc[:8], dim=1).values, rtol=0, atol=2e-6)
    # THE ORACLE at full size: 64 queries of the batch (every 4th, so all eight query-owning waves are
    # covered) against search_prepared on the bf16-rounded corpus and the device-prepared queries — ids
    # compared exactly wherever the f64 decision gaps exceed 1e-6, scores to 1e-5 (tests/parity.py)
    Eb = cb.emb.float().cpu().numpy()
    sel = list(range(0, 256, 4))
    Qp = device_prepared_queries(Qb[sel].cpu().numpy())
    ih, sh_ = ids.cpu().numpy(), sc.cpu().numpy()
    n_dec = 0
    for t, j in enumerate(sel):
        decisive, msg = compare_query(Eb, Qp[t], dewi32, ent32, k, eta, 0.0, "cosine", ih[j], sh_[j], exact_gaps=False,
                                      gap=1e-6, score_tol=1e-5, prepared=True)
        assert msg is None, (j, msg)
        n_dec += int(decisive)
    assert n_dec >= 48, n_dec            # k = 100 leaves ~10 % of queries with an adjacent gap under 1e-6


def test_c5_row_cosine_full_size():
    """Config C5's I_hat kernel (row_cosine_512_kernel) at its full size, 1M x 512 x 2 matrices, against
    torch's CPU F.cosine_similarity (the reference's own call, signals/cross_modal.py:69)."""
    import torch
    from dewi import signals
    n, d = 1_000_000, 512
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    A = torch.randn((n, d), generator=g, device="cuda")
    g.manual_seed(43)
    B = torch.randn((n, d), generator=g, device="cuda")
    B[::7] = A[::7] * 0.5 + B[::7] * 0.05          # a share of strongly aligned pairs, not only ~0 cosines
    A[123] = 0                                      # zero row: eps clamp, as torch
    got = signals.cross_modal_similarity(A, B)
    got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
    want = torch.nn.functional.cosine_similarity(A.cpu(), B.cpu()).numpy()
    assert got.shape == (n,)
    assert np.max(np.abs(got - want)) <= 2e-6
    assert got[123] == 0.0


def test_c4_eight_shards_on_one_gpu():
    """BASELINE configs[3] replayed on ONE GPU: eight resident 1M x 768 fp32 shards (24.6 GB), each scanned by
    dewi_knn_candidates with its id_offset, records stacked and merged by dewi_merge_rerank — the whole
    exchange minus the wire.  Must equal (i) ONE search over the 8M-row matrix bit for bit and (ii) the oracle
    on 8 queries.  Multi-GPU proper (RCCL with > 1 rank) is NOT measured by this test."""
    import torch
    from dewi import _engine as eng
    from dewi import _native as nat
    S, n, d, k, eta = 8, 1_000_000, 768, 10, 0.3
    total = S * n
    big = torch.empty((total, d), dtype=torch.float32, device="cuda")
    for s in range(S):                               # seeds 42..49
        g = torch.Generator(device="cuda")
        g.manual_seed(42 + s)
        for r0 in range(0, n, 250_000):
            big[s * n + r0: s * n + r0 + 250_000] = torch.randn((250_000, d), generator=g, device="cuda")
    nat.check(nat.load_library().dewi_normalize_rows_f32(nat.ptr(big), nat.ptr(big), total, d, nat.stream_ptr()))
    cols = orc.synth_payload_columns(total, seed=42)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    dd, ee = torch.from_numpy(dewi32).cuda(), torch.from_numpy(ent32).cuda()
    whole = eng.DeviceCorpus(big, dd, ee, "cosine")
    shards = [eng.DeviceCorpus(big[s * n:(s + 1) * n], dd[s * n:(s + 1) * n], ee[s * n:(s + 1) * n], "cosine",
                               id_offset=s * n) for s in range(S)]
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    Q = torch.randn((16, d), generator=g, device="cuda")
    c = 2 * k
    lists = torch.stack([sh.candidates_device(Q, c) for sh in shards])           # [8, 16, 20, 4]
    recs = eng.records_to_numpy(lists)
    for s in range(S):                                                           # global ids, in range per shard
        assert recs["id"][s].min() >= s * n and recs["id"][s].max() < (s + 1) * n
    ids, sc = eng.merge_rerank_device(lists, c, k, eta, 0.0)
    ids_w, sc_w = whole.search_device(Q, k, eta, 0.0)
    assert torch.equal(ids, ids_w) and torch.equal(sc, sc_w)
    assert ids.max().item() >= n                                                 # answers do come from later shards
    E = big.cpu().numpy()
    Qh, ih, sh_ = Q.cpu().numpy(), ids.cpu().numpy(), sc.cpu().numpy()
    n_dec = 0
    for j in range(8):
        decisive, msg = compare_query(E, Qh[j], dewi32, ent32, k, eta, 0.0, "cosine", ih[j], sh_[j], exact_gaps=False)
        assert msg is None, (j, msg)
        n_dec += int(decisive)
    assert n_dec >= 6
    # the same exchange at k = 200 and k = 1500: 8 x 400 and 8 x 3000 records per query — beyond the 2048 an LDS sort
    # holds, so the shard select (k = 1500) and the merge (both) run through their global-memory forms.  The sharded
    # answer must still be the single-device answer bit for bit, and the oracle's (near-tie rules at such k).
    for k2 in (200, 1500):
        c2 = 2 * k2
        lists2 = torch.stack([sh.candidates_device(Q[:2], c2) for sh in shards])
        ids2, sc2 = eng.merge_rerank_device(lists2, c2, k2, eta, 0.0)
        ids_w2, sc_w2 = whole.search_device(Q[:2], k2, eta, 0.0)
        assert torch.equal(ids2, ids_w2) and torch.equal(sc2, sc_w2), k2
        _, msg = compare_query(E, Qh[0], dewi32, ent32, k2, eta, 0.0, "cosine", ids2[0].cpu().numpy(), sc2[0].cpu().numpy(),
                               exact_gaps=False)
        assert msg is None, (k2, msg)
