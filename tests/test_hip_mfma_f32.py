"""GPU parity tests of the batched fp32 matrix-core path (csrc/knn_mfma_f32.hip) through the C ABI.

The path is taken for >= 5 queries over an fp32 corpus of >= 64 K rows with dim % 256 == 0 (<= 1536),
cosine space (l2: dims to 768, and bf16 corpora of any supported dim), at most 256 candidates.  Up to dim 1024 every fp32 value is cut into three bf16 pieces
(x = hi + mid + lo exactly) and a block of products is six bf16 matrix instructions with exact products
and fp32 accumulation; the three dropped cross terms are below 2^-23 of |q_i e_i| each, the size of one
fp32 rounding (dim 1536 keeps v_mfma_f32_32x32x2_f32 on the fp32 values).  So the oracle comparison is
the plain fp32 one (tests/parity.py: ids exactly wherever the f64 decision gaps exceed 5e-7, scores to
1e-5), and ``test_mfma_f32_heavy_tailed_components`` bounds the score error itself on vectors whose
components span six orders of magnitude.  ``search_device`` is called, so an unanswered query (-1) would be
seen, not repaired.
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import check_batch, device_prepared_queries

pytestmark = pytest.mark.gpu

# The UNREFINED form of space="l2" on the matrix cores is an opt-in (dewi_tuning_set batched_mfma = 2: its score
# 2<e,q> - ||e||^2 - ||q||^2 has an absolute error of ~ulp(||e||^2 + ||q||^2)).  By default l2 batches over an fp32 corpus
# take the pass in exact-refine mode (bit-equal to the one-query search; tests/test_hip_round3.py checks that mode against
# the oracle and against the one-query search) and l2 batches over a bf16 corpus the exact row kernels.
# This module tests the matrix-core passes themselves, unrefined, in both spaces: every test runs opted in.
MFMA_ON = 2


@pytest.fixture(autouse=True)
def _l2_on_the_matrix_cores():
    from dewi import _engine as eng
    eng.tuning(0, 0, -1, MFMA_ON)
    yield
    eng.tuning(0, 0, -1, 1)


def _corpus(n, dim, seed):
    from dewi import _engine as eng
    raw = orc.synth_corpus(n, dim, seed=seed)
    cols = orc.synth_payload_columns(n, seed=seed)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    return c, c.emb.cpu().numpy(), dewi32, ent32


@pytest.mark.parametrize("dim,n,b,k", [(768, 70_001, 32, 10), (768, 66_000, 5, 10), (512, 80_000, 70, 10),
                                       (256, 131_073, 33, 100), (1024, 65_536, 8, 10), (1536, 65_600, 17, 10),
                                       (768, 300_000, 64, 128),
                                       # round 4: a PARTIAL last chunk — rows of whole 16-byte units up to 1536 columns take the pass
                                       (384, 70_001, 32, 10), (640, 66_000, 12, 10), (96, 100_000, 7, 10), (992, 65_600, 33, 10),
                                       (160, 70_000, 64, 100), (32, 66_000, 5, 10),
                                       # ... and 1280 / 2048 columns (whole chunks, fp32 instruction as at 1536), 1312 = five chunks + one wave
                                       (1280, 66_000, 12, 10), (1312, 65_600, 7, 10), (2048, 65_600, 8, 10),
                                       # ... and rows that END INSIDE a wave's 32-column slice (dim % 4 == 0, not % 32): 300 (12 columns
                                       # into wave 1), 100, 1000, 36, 260 / 1284 (one unit into a new chunk), 252 (wave 7 four short)
                                       (300, 70_001, 32, 10), (100, 100_000, 7, 10), (1000, 65_600, 33, 10), (36, 66_000, 5, 10),
                                       (260, 70_000, 12, 10), (1284, 65_600, 6, 10), (252, 66_000, 40, 50)])
def test_mfma_f32_batched_vs_oracle(dim, n, b, k):
    import torch
    from dewi import _engine as eng
    c, E, dewi32, ent32 = _corpus(n, dim, seed=dim + b)
    Q = orc.synth_queries(b, dim, seed=b)
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, 0.3, 0.1)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0 and not np.isnan(sc).any()             # no query was refused (nothing repaired here)
    check_batch(E, Q, dewi32, ent32, k, 0.3, 0.1, "cosine", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.5,
                exact_gaps=False)
    # deterministic, and the same rows as the row-per-wave kernels (matrix-core path switched off): the two paths
    # sum in different orders, so rare near-tie swaps only
    ids2, sc2 = c.search_device(torch.from_numpy(Q).cuda(), k, 0.3, 0.1)
    assert np.array_equal(ids2.cpu().numpy(), ids) and np.array_equal(sc2.cpu().numpy(), sc)
    eng.tuning(0, 0, -1, 0)
    try:
        ids_s, sc_s = c.search(Q[:8], k, 0.3, 0.1)
    finally:
        eng.tuning(0, 0, -1, MFMA_ON)
    assert np.mean(ids_s == ids[:8]) > 0.98
    assert np.allclose(np.sort(sc_s, axis=1), np.sort(sc[:8], axis=1), rtol=0, atol=2e-6)


def test_mfma_f32_heavy_tailed_components():
    """Components spanning six orders of magnitude, a few of them carrying the norm: the three-piece cut of the
    fp32 values must still give fp32-grade scores (|score - f64 cosine| <= 3e-7, the error of a plain fp32 chain
    at this depth), not bf16-grade ones (4e-3) or two-piece ones (1.5e-5)."""
    import torch
    from dewi import _engine as eng
    n, dim, b, k = 70_000, 768, 32, 10
    rng = np.random.default_rng(5)
    raw = (rng.standard_normal((n, dim)) * np.exp(3.0 * rng.standard_normal((n, dim)))).astype(np.float32)
    Q = (rng.standard_normal((b, dim)) * np.exp(3.0 * rng.standard_normal((b, dim)))).astype(np.float32)
    Q[:8] = raw[1000:1008] * (1.0 + 1e-3 * rng.standard_normal((8, dim)).astype(np.float32))   # near-duplicates: cosines near 1
    cols = orc.synth_payload_columns(n, seed=5)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    E = c.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, 0.0, 0.0)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0
    assert [int(ids[j, 0]) for j in range(8)] == list(range(1000, 1008))
    qn = Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True)
    exact = np.einsum("bkd,bd->bk", E[ids].astype(np.float64), qn)
    assert np.max(np.abs(exact - sc)) <= 3e-7, np.max(np.abs(exact - sc))
    check_batch(E, Q, dewi32, ent32, k, 0.0, 0.0, "cosine", ids, sc, min_decisive_frac=0.5, exact_gaps=False)


def test_mfma_f32_a_batch_of_four_stays_on_the_scan_kernels_and_agrees():
    """Batches of 2-4 queries keep the row-per-wave kernels (one pass for four queries is as fast); the answers of
    the two paths for the same queries must be interchangeable."""
    import torch
    c, E, dewi32, ent32 = _corpus(70_000, 768, seed=11)
    Q = orc.synth_queries(12, 768, seed=12)
    ids12, sc12 = c.search(Q, 10, 0.3, 0.0)                       # matrix-core path
    ids4 = np.concatenate([c.search(Q[i:i + 4], 10, 0.3, 0.0)[0] for i in (0, 4, 8)])
    assert np.mean(ids4 == ids12) > 0.98
    check_batch(E, Q, dewi32, ent32, 10, 0.3, 0.0, "cosine", ids12, sc12, exact_gaps=False)


@pytest.mark.parametrize("dim,bf16", [(384, False), (384, True), (96, False), (96, True), (800, False), (800, True),
                                       # rows that end inside a wave's slice: the units behind the end are cleared in registers
                                       (200, False), (200, True), (1000, False), (1000, True), (300, False), (100, False), (260, False),
                                       (1284, False)])
def test_partial_chunk_keeps_a_nan_row_out_of_its_neighbours(dim, bf16):
    """dim % 256 != 0: behind the end of a row's partial last chunk the LDS ring holds whatever an earlier chunk left there
    (the DMA lanes behind the end move nothing); the waves whose columns lie past the row skip their matrix instructions and
    the wave the row ends in clears the units behind the end, so a NaN row (a zero embedding: 0/0, as in the reference) must
    not turn any other row into NaN.  The NaN row itself ranks first for every query (NumPy's partition order)."""
    from dewi import _engine as eng
    import torch
    n, k, b = 70_000, 5, 16
    raw = orc.synth_corpus(n, dim, seed=dim)
    bad = [1000, 31 + 32 * 7, n - 1]                   # mid-tile, last row of a tile, last row of the corpus
    for i in bad:
        raw[i] = 0.0
    cols = orc.synth_payload_columns(n, seed=dim)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    if bf16:
        c = c.to_bf16()
    assert c.scan_kernel_name(b, k).startswith("mfma_scan_f32")
    Q = orc.synth_queries(b, dim, seed=3)
    Q[0], Q[1], Q[2] = raw[999], raw[30 + 32 * 7], raw[n - 2]        # the rows stored right in front of the NaN rows
    ids, sc = (t.cpu().numpy() for t in c.search_device(torch.from_numpy(Q).cuda(), k, 0.0, 0.0))
    assert all(sorted(row[:3].tolist()) == sorted(bad) for row in ids) and np.isnan(sc[:, :3]).all()   # NaN rows first
    assert not np.isnan(sc[:, 3:]).any()
    assert ids[0, 3] == 999 and ids[1, 3] == 30 + 32 * 7 and ids[2, 3] == n - 2
    assert np.allclose(sc[:3, 3], 1.0, atol=1e-2 if bf16 else 1e-5)
    one = [c.search_device(torch.from_numpy(Q[j:j + 1]).cuda(), k, 0.0, 0.0)[0].cpu().numpy()[0] for j in range(b)]
    assert np.mean(np.stack(one) == ids) > 0.95        # the row kernels agree up to near-tie swaps


def test_mfma_f32_overflow_is_repaired_behind_the_c_abi():
    """40 000 exact duplicates of a query's best document overflow that query's survivor segments: the depth-split pass
    refuses it and the repair launches of the same call answer it on the fp32 row kernels (raw ABI call, no host repair)."""
    from dewi import _engine as eng
    import torch
    n, dim, k = 100_000, 256, 10
    raw = orc.synth_corpus(n, dim, seed=3)
    raw[50_000:90_000] = raw[7]
    cols = orc.synth_payload_columns(n, seed=3)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = orc.synth_queries(16, dim, seed=4)
    Q[5] = raw[7]
    q_dev = torch.from_numpy(Q).cuda()
    ids_raw, sc_raw = c.search_device(q_dev, k, 0.0, 0.0)
    assert c.refused_by_last_call().tolist() == [j == 5 for j in range(16)]
    ids_raw, sc_raw = ids_raw.cpu().numpy(), sc_raw.cpu().numpy()
    assert ids_raw.min() >= 0
    one_ids, one_sc = c.search_device(q_dev[5:6].contiguous(), k, 0.0, 0.0)
    assert np.array_equal(ids_raw[5], one_ids.cpu().numpy()[0]) and np.array_equal(sc_raw[5], one_sc.cpu().numpy()[0])
    assert ids_raw[5].tolist() == [7] + list(range(50_000, 50_009))
    assert np.allclose(sc_raw[5], 1.0, atol=1e-5)


@pytest.mark.parametrize("space", ["cosine", "l2"])
def test_mfma_f32_shard_candidates_and_merge_equal_whole(space):
    """Two fp32 shards answered by the matrix-core path (dewi_knn_candidates) + merge == the whole corpus."""
    import torch
    from dewi import _engine as eng
    n, d, k = 140_000, 256, 10
    raw = orc.synth_corpus(n, d, seed=21) * np.float32(1.0 if space == "cosine" else 1.7)
    cols = orc.synth_payload_columns(n, seed=21)
    Q = torch.from_numpy(orc.synth_queries(20, d, seed=22) * np.float32(1.0 if space == "cosine" else 0.1)).cuda()
    whole = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space=space)
    ids_w, sc_w = whole.search_device(Q, k, 0.3, 0.2)
    lists = []
    for lo, hi in ((0, 70_000), (70_000, n)):
        sub = {key: v[lo:hi] for key, v in cols.items()}
        sh = eng.DeviceCorpus.from_host(raw[lo:hi], sub["dewi"], sub["ht_mean"], sub["hi_mean"], id_offset=lo, space=space)
        lists.append(sh.candidates_device(Q, 2 * k))
    ids, sc = eng.merge_rerank_device(torch.stack(lists), 2 * k, k, 0.3, 0.2)
    # per-row sums are the same whichever shard a row is in (same kernel, same depth split), so bit-equal
    assert torch.equal(ids, ids_w) and torch.equal(sc, sc_w)


@pytest.mark.parametrize("bf16,dim,n,b,k", [(False, 768, 70_001, 32, 10), (False, 256, 131_073, 12, 10), (False, 512, 80_000, 40, 50),
                                            (True, 768, 70_001, 32, 10), (True, 1024, 65_600, 9, 10), (True, 256, 100_000, 70, 100)])
def test_mfma_l2_batched_vs_oracle(bf16, dim, n, b, k):
    """space="l2" (reference backends.py:434-436: -sum((E - q)^2)) on the depth-split matrix-core pass, which scores
    2<e,q> - ||e||^2 - ||q||^2 with the row norms summed in the kernel from the fragments it multiplies.  Rows and
    queries of very different lengths (norms 0.5 .. 2), so the norm terms decide the ranking as much as the products.
    fp32 corpus: plain oracle comparison (gaps and tolerance scaled by the score magnitude, tests/parity.py); bf16
    corpus: the oracle on the stored bf16 rows and the device's bf16-rounded queries.  70 queries over a bf16 corpus:
    the 256-query kernel is cosine-only, so the batch runs as three depth-split passes."""
    import torch
    from dewi import _engine as eng
    rng = np.random.default_rng(dim + b)
    raw = orc.synth_corpus(n, dim, seed=dim + b) * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    Q = orc.synth_queries(b, dim, seed=b) * rng.uniform(0.5, 2.0, size=(b, 1)).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim + b)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space="l2")
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    kw = dict(exact_gaps=False)
    if bf16:
        c = c.to_bf16()
        E = c.emb.float().cpu().numpy()
        Qo = device_prepared_queries(Q, "l2")
        kw = dict(gap=1e-6, score_tol=1e-5, prepared=True)
    else:
        E, Qo = c.emb.cpu().numpy(), Q
        assert np.array_equal(E, raw)                                   # l2 keeps raw rows
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, 0.3, 0.1)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0 and not np.isnan(sc).any()
    # (floor 0.2 at k = 100: the harness scales the cut gap by |similarity at the cut| — hundreds in l2 — since round 3)
    check_batch(E, Qo, dewi32, ent32, k, 0.3, 0.1, "l2", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.2, **kw)
    # the same rows as the exact row-per-wave l2 kernels (matrix-core paths switched off), up to near-tie swaps
    eng.tuning(0, 0, -1, 0)
    try:
        ids_s, sc_s = c.search(Q[:8], k, 0.3, 0.1)
    finally:
        eng.tuning(0, 0, -1, MFMA_ON)
    assert np.mean(ids_s == ids[:8]) > 0.97
    assert np.allclose(np.sort(sc_s, axis=1), np.sort(sc[:8], axis=1), rtol=1e-6, atol=1e-5)    # scores are -||e - q||^2: hundreds


@pytest.mark.parametrize("space,bf16", [("cosine", False), ("l2", False), ("l2", True)])
def test_mfma_depth_pass_edge_rows(space, bf16):
    """Edge rows on the depth-split pass, against the exact row-per-wave kernels (matrix-core paths off) of the same
    library, which the small-size tests pin to the oracle: a NaN row ranks first for every query (NumPy's partition
    order, reference backends.py:439-444), queries that ARE corpus rows find their row first — under l2 with a score
    that is zero up to the rounding of 2<e,q> - ||e||^2 - ||q||^2 at the magnitude of ||e||^2 —, an all-zero row and a
    partial last tile change nothing."""
    import torch
    from dewi import _engine as eng
    n, dim, b, k = 70_003, 512, 16, 10
    rng = np.random.default_rng(11)
    raw = orc.synth_corpus(n, dim, seed=11) * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    raw[40_000] = np.nan if space == "l2" else 0.0        # cosine: a zero row is NaN after the build's normalisation
    raw[123] = 0.0 if space == "l2" else raw[123]
    Q = orc.synth_queries(b, dim, seed=12) * np.float32(0.05)
    Q[:4] = raw[[5, 69_999, 70_002, 31_000]]                # rows of the first, a middle and the (partial) last tile
    cols = orc.synth_payload_columns(n, seed=11)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space=space)
    if bf16:
        c = c.to_bf16()
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, 0.0, 0.0)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert (ids[:, 0] == 40_000).all() and np.isnan(sc[:, 0]).all() and not np.isnan(sc[:, 1:]).any()
    assert ids[:4, 1].tolist() == [5, 69_999, 70_002, 31_000]
    if space == "l2":
        own = np.sum(raw[[5, 69_999, 70_002, 31_000]].astype(np.float64) ** 2, axis=1)
        assert np.all(np.abs(sc[:4, 1]) <= (4e-3 if bf16 else 2e-6) * own + 1e-6), sc[:4, 1]
    eng.tuning(0, 0, -1, 0)
    try:
        ids_s, sc_s = c.search(Q, k, 0.0, 0.0)
    finally:
        eng.tuning(0, 0, -1, MFMA_ON)
    assert np.mean(ids_s == ids) > 0.98
    assert np.allclose(np.sort(sc_s[:, 1:], axis=1), np.sort(sc[:, 1:], axis=1), rtol=2e-6, atol=2e-5 if space == "l2" else 2e-6)


def _random_depth_cases(n_cases=14, seed=2024):
    rs = np.random.RandomState(seed)
    cases = []
    for i in range(n_cases):
        bf16 = bool(rs.randint(2))
        space = "l2" if rs.rand() < 0.4 else "cosine"
        dims = [256, 512, 768, 1024, 1536] if (bf16 or space == "cosine") else [256, 512, 768]
        dim = int(rs.choice(dims))
        n = int(rs.randint(65_536, 110_000))
        b = int(rs.randint(5 if not bf16 else 2, 71))
        if bf16 and space == "cosine" and dim <= 768 and b > 32:
            b = int(rs.randint(2, 33))                     # larger bf16 cosine batches belong to the 256-query kernel's tests
        k = int(rs.choice([1, 5, 10, 50, 128]))
        cases.append((i, bf16, space, dim, n, b, k, float(rs.choice([0.0, 0.3, 0.7])), float(rs.choice([0.0, 0.2]))))
    return cases


@pytest.mark.parametrize("case", _random_depth_cases(), ids=lambda c: f"{c[0]}-{'bf16' if c[1] else 'f32'}-{c[2]}-d{c[3]}-n{c[4]}-b{c[5]}-k{c[6]}")
def test_depth_pass_randomised_cases(case):
    """Seeded random shapes on the depth-split pass — ragged last tiles (n % 32), partial query groups (b % 32), one to
    three passes, k from 1 to 128 (2k candidates up to 256), both spaces, both element types, every chunk count — each
    against the oracle (bf16: on the stored rows and the device's prepared queries)."""
    import torch
    from dewi import _engine as eng
    i, bf16, space, dim, n, b, k, eta, pref = case
    rng = np.random.default_rng(1000 + i)
    raw = orc.synth_corpus(n, dim, seed=300 + i)
    Q = orc.synth_queries(b, dim, seed=400 + i)
    if space == "l2":
        raw = raw * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
        Q = Q * np.float32(0.05) * rng.uniform(0.5, 2.0, size=(b, 1)).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=300 + i)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space=space)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    kw = dict(exact_gaps=False)
    if bf16:
        c = c.to_bf16()
        E, Qo = c.emb.float().cpu().numpy(), device_prepared_queries(Q, space)
        kw = dict(gap=1e-6, score_tol=1e-5, prepared=True)
    else:
        E, Qo = c.emb.cpu().numpy(), Q
    ids_d, sc_d = c.search_device(torch.from_numpy(Q).cuda(), k, eta, pref)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0 and not np.isnan(sc).any()
    check_batch(E, Qo, dewi32, ent32, k, eta, pref, space, ids, sc, min_decisive_frac=0.6 if k <= 10 else 0.25, **kw)
