"""GPU parity tests of the batched bf16 matrix-core path (config C3's kernel) through the C ABI.

The path is taken for >= 2 queries over a bf16 corpus of >= 64 K rows with dim % 128 == 0.
Parity is checked in two exact steps so that the id ranking can be compared at fp32-summation
noise (gap 1e-6) instead of at a bf16 ulp:

  1. query preparation alone (``dewi_prepare_queries_bf16``, the kernel the path runs on its
     queries): every element within one bf16 ulp of the oracle's ``bf16_round(prepare_query(q))``
     and at least 99 % of them bit-equal (the GPU and NumPy sum ||q||^2 in different orders, so an
     element on a rounding boundary may round the other way);
  2. the search against the oracle run on THOSE prepared queries and the bf16-rounded corpus:
     products of bf16 values are exact in fp32, so what is left is the accumulation order.

``search_device`` is called (not the blocking ``search``), so an unanswered query (id -1) would be
seen here, not repaired.  The batched path and the small-batch scan kernels must also agree with
each other.
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
TOL = dict(gap=1e-6, score_tol=1e-5, prepared=True, exact_gaps=False)


def _corpus(n, dim, seed, space="cosine"):
    from dewi import _engine as eng
    raw = orc.synth_corpus(n, dim, seed=seed)
    cols = orc.synth_payload_columns(n, seed=seed)
    cb = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space).to_bf16()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    return cb, cb.emb.float().cpu().numpy(), dewi32, ent32


@pytest.mark.parametrize("dim,n,b,k", [(768, 70_001, 256, 100), (768, 66_000, 40, 10), (512, 80_000, 300, 10),
                                       (256, 70_000, 17, 100), (128, 131_072, 64, 10), (128, 1_100_000, 32, 10),
                                       # up to 32 queries, and dimensions the 256-query kernel cannot hold in
                                       # registers: the depth-split pass (csrc/knn_mfma_f32.hip, bf16 geometry)
                                       (768, 70_001, 8, 10), (768, 70_001, 32, 100), (512, 80_000, 2, 10),
                                       (1024, 65_600, 70, 10), (1536, 66_000, 40, 10),
                                       # round 4: the depth-split pass with a PARTIAL last chunk (dim % 32 == 0, not % 256)
                                       (384, 70_000, 8, 10), (96, 131_072, 12, 10), (640, 66_000, 32, 10), (992, 65_600, 5, 10),
                                       (160, 70_001, 70, 100),
                                       # ... and the wider whole-chunk rows: 1280, 2048, 3072, 4096 (bf16: 8 query registers per chunk)
                                       (1280, 65_600, 9, 10), (1312, 65_600, 33, 10), (2048, 65_600, 4, 10), (3072, 65_536, 12, 10),
                                       (4096, 65_536, 32, 10)])
def test_mfma_batched_vs_oracle(dim, n, b, k):
    import torch
    cb, Eb, dewi32, ent32 = _corpus(n, dim, seed=dim + b)
    Q = orc.synth_queries(b, dim, seed=b)
    Qp = device_prepared_queries(Q)
    ids_d, sc_d = cb.search_device(torch.from_numpy(Q).cuda(), k, 0.3, 0.1)
    ids, sc = ids_d.cpu().numpy(), sc_d.cpu().numpy()
    assert ids.min() >= 0 and not np.isnan(sc).any()             # no query was refused (nothing repaired here)
    check_batch(Eb, Qp, dewi32, ent32, k, 0.3, 0.1, "cosine", ids, sc, min_decisive_frac=0.75, **TOL)
    # same answers as the small-batch kernels (matrix-core path switched off for the comparison)
    from dewi import _engine as eng
    eng.tuning(0, 0, -1, 0)
    try:
        ids_s = np.concatenate([cb.search(Q[i:i + 4], k, 0.3, 0.1)[0] for i in range(0, min(b, 32), 4)])
    finally:
        eng.tuning(0, 0, -1, MFMA_ON)
    agree = np.mean(ids_s == ids[: ids_s.shape[0]])
    assert agree > 0.98, agree          # the two paths sum in different orders: rare near-tie swaps only


def test_mfma_overflow_is_repaired_behind_the_c_abi():
    """40 000 exact duplicates of the query's best document overflow the candidate buffer of that query (capacity
    4*32*c): the matrix-core pass refuses it, and the repair launches of the SAME library call answer it on the exact row
    kernels — the raw ABI call (search_device: no host look at the result) returns what the one-query search returns."""
    from dewi import _engine as eng
    import torch
    n, dim, k = 100_000, 256, 10
    raw = orc.synth_corpus(n, dim, seed=3)
    raw[50_000:90_000] = raw[7]                       # rows 50000..89999 == row 7
    cols = orc.synth_payload_columns(n, seed=3)
    cb = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"]).to_bf16()
    Q = orc.synth_queries(32, dim, seed=4)
    Q[5] = raw[7]                                     # query 5 hits the duplicated document
    q_dev = torch.from_numpy(Q).cuda()
    ids_raw, sc_raw = cb.search_device(q_dev, k, 0.0, 0.0)
    assert cb.refused_by_last_call().tolist() == [j == 5 for j in range(32)]      # the pass really refused it (and only it)
    ids_raw, sc_raw = ids_raw.cpu().numpy(), sc_raw.cpu().numpy()
    assert ids_raw.min() >= 0 and not np.isnan(sc_raw).any()
    one_ids, one_sc = cb.search_device(q_dev[5:6].contiguous(), k, 0.0, 0.0)      # one query: row kernels
    assert not cb.refused_by_last_call().any()
    assert np.array_equal(ids_raw[5], one_ids.cpu().numpy()[0]) and np.array_equal(sc_raw[5], one_sc.cpu().numpy()[0])
    ids, sc = cb.search(Q, k, 0.0, 0.0)               # the blocking API is the same call
    assert np.array_equal(ids, ids_raw) and np.array_equal(sc, sc_raw)
    assert ids[5].tolist() == [7] + list(range(50_000, 50_009))      # ties: lower rows first
    assert np.allclose(sc[5], 1.0, atol=1e-2)


@pytest.mark.parametrize("space", ["cosine", "l2"])
def test_pipelined_batches_in_both_spaces(space):
    """dewi_knn_finish plans from the space it is given (ABI 3): an l2 batch of 12 over an fp32 corpus is a depth-split
    pass, over a 40 000-row corpus the row kernels — scan and finish must agree on that in either space."""
    from dewi import _engine as eng
    import torch
    for n in (100_000, 40_000):
        dim, b, k = 256, 12, 10
        raw = orc.synth_corpus(n, dim, seed=5) * np.float32(1.5)
        cols = orc.synth_payload_columns(n, seed=5)
        c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space=space)
        q_dev = torch.from_numpy(orc.synth_queries(b, dim, seed=6) * np.float32(0.1)).cuda()
        want_ids, want_sc = c.search_device(q_dev, k, 0.3, 0.1)
        pipe = eng.PipelinedSearcher(c, k, 0.3, 0.1, n_queries=b)
        ids = torch.empty((b, k), dtype=torch.int64, device="cuda")
        sc = torch.empty((b, k), dtype=torch.float32, device="cuda")
        pipe.submit(q_dev, ids, sc)
        pipe.drain()
        assert torch.equal(ids, want_ids) and torch.equal(sc, want_sc)


@pytest.mark.parametrize("bf16,b", [(True, 40), (True, 8), (False, 12)])
def test_pipelined_batches_take_the_matrix_core_paths_and_equal_the_one_call_search(bf16, b):
    """dewi_knn_scan + dewi_knn_finish (two streams, rotating workspaces) choose the same path as the one-call
    search for a query batch — 256-query kernel, depth-split pass over a bf16 / an fp32 corpus — so the answers are
    bit-equal; a refused query of a pipelined batch is repaired inside ``dewi_knn_finish`` (final results and shard
    records alike: no marker leaves the library)."""
    from dewi import _engine as eng
    import torch
    n, dim, k = 100_000, 256, 10
    raw = orc.synth_corpus(n, dim, seed=3)
    raw[50_000:90_000] = raw[7]
    cols = orc.synth_payload_columns(n, seed=3)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    if bf16:
        c = c.to_bf16()
    Q = orc.synth_queries(3 * b, dim, seed=4).reshape(3, b, dim)
    Q[1, 5] = raw[7]                                  # batch 1, query 5 overflows its survivor segments
    q_dev = torch.from_numpy(Q).cuda()
    want = [c.search(Q[i], k, 0.3, 0.1) for i in range(3)]
    raw_ids, _ = c.search_device(q_dev[1], k, 0.3, 0.1)
    assert c.refused_by_last_call().tolist() == [j == 5 for j in range(b)]     # the pass refuses it, the call answers it
    assert np.array_equal(raw_ids.cpu().numpy(), want[1][0])
    pipe = eng.PipelinedSearcher(c, k, 0.3, 0.1, n_queries=b, depth=3, scan_streams=2)
    ids = torch.empty((3, b, k), dtype=torch.int64, device="cuda")
    sc = torch.empty((3, b, k), dtype=torch.float32, device="cuda")
    for rounds in range(2):                                         # workspaces and output buffers reused
        for i in range(3):
            pipe.submit(q_dev[i], ids[i], sc[i])
    pipe.drain()
    for i in range(3):
        assert np.array_equal(ids[i].cpu().numpy(), want[i][0]) and np.array_equal(sc[i].cpu().numpy(), want[i][1])
    recs = torch.empty((b, 2 * k, 4), dtype=torch.int32, device="cuda")
    pipe2 = eng.PipelinedSearcher(c, k, 0.3, 0.1, n_queries=b, n_candidates=2 * k)
    pipe2.submit(q_dev[1], out_records=recs)
    pipe2.drain()
    assert torch.equal(recs, c.candidates_device(q_dev[1], 2 * k))
    assert (recs[:, :, 3] >= 0).all().item()                        # no refusal marker: query 5's records are real
    assert torch.equal(recs[5], c.candidates_device(q_dev[1, 5:6].contiguous(), 2 * k)[0])   # == the row kernels' records
