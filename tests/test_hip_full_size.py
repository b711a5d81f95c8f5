"""Full-size checks at BASELINE.json configs[1] (1M x 768 fp32, k=10, eta=0.3) and configs[2]
(bf16, 256 queries, k=100): properties that do not need the oracle on every query — sortedness,
determinism, shard-independence (3 ragged shards + merge == whole) — plus the oracle itself on a
handful of queries.
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import compare_query

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
    ids_b, sc_b = c.search_device(Q, k, eta, 0.0)                               # NQ=4 passes == NQ=1 passes
    assert torch.equal(ids_b, ids) and torch.equal(sc_b, sc)
    ids_again, sc_again = c.search_device(Q, k, eta, 0.0)                       # deterministic
    assert torch.equal(ids_again, ids_b) and torch.equal(sc_again, sc_b)
    # shard-independence: 3 ragged shards (views of the same matrix) + merge == whole
    bounds = [0, 333_333, 700_001, c.n_rows]
    lists = []
    for s in range(3):
        lo, hi = bounds[s], bounds[s + 1]
        sh = eng.DeviceCorpus(c.emb[lo:hi], c.dewi32[lo:hi], c.ent32[lo:hi], "cosine", id_offset=lo)
        lists.append(sh.candidates_device(Q, 2 * k))
    m_ids, m_sc = eng.merge_rerank_device(torch.stack(lists), 2 * k, k, eta, 0.0)
    assert torch.equal(m_ids, ids) and torch.equal(m_sc, sc)
    # the oracle on 8 of the queries
    E = c.emb.cpu().numpy()
    Qh, ih, sh_ = Q.cpu().numpy(), ids.cpu().numpy(), sc.cpu().numpy()
    for j in range(8):
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
    # ... and a batch of 8 takes the matrix-core path too: same answers as inside the batch of 256
    ids_8, sc_8 = cb.search_device(Qb[:8].contiguous(), k, eta, 0.0)
    assert torch.equal(ids_8, ids[:8]) and torch.equal(sc_8, sc[:8])
