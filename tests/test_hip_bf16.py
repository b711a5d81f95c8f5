"""GPU parity tests of the bf16-corpus search (config C3's storage) through the C ABI.

Oracle: the fp32 restatement of the reference run on bf16-ROUNDED inputs (products of bf16 values
are exact in fp32).  The GPU and NumPy normalise the query with differently computed norms, so a query
element that sits on a bf16 rounding boundary can round the other way (one bf16 ulp on ONE of the d
products, <= ~5e-6 on a unit-norm score).  To keep that out of the ranking comparison the tests check the
query preparation on its own (``parity.device_prepared_queries``: within one bf16 ulp of the oracle's,
>= 99 % bit-equal) and then run the oracle on the device-prepared queries: what is left is fp32 summation
order, so ids are compared exactly wherever decision gaps exceed 1e-6 and scores to 1e-5.
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import check_batch, device_prepared_queries

pytestmark = pytest.mark.gpu
TOL = dict(gap=1e-6, score_tol=1e-5, prepared=True)


def _setup(n, dim, seed):
    from dewi import _engine as eng
    raw = orc.synth_corpus(n, dim, seed=seed)
    cols = orc.synth_payload_columns(n, seed=seed)
    c32 = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    cb = c32.to_bf16()
    Eb = cb.emb.float().cpu().numpy()
    # device rounding == oracle rounding of the same stored fp32 rows
    assert np.array_equal(Eb, orc.bf16_round(c32.emb.cpu().numpy()))
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    return cb, Eb, dewi32, ent32


@pytest.mark.parametrize("dim,n", [(768, 4001), (768, 4000), (512, 3000), (256, 2501), (1024, 1501), (136, 2000),
                                   (100, 2000), (1280, 900), (8, 3000), (64, 3000), (200, 2000), (264, 2001), (384, 2001),
                                   (1000, 1501), (3072, 800), (5120, 502), (8192, 301), (8200, 300)])
def test_bf16_search_vs_oracle(dim, n):
    """Pair-of-rows kernel (dim = 256*H, odd and even row counts); the any-width kernels (dim % 8 == 0: rows sharing a wave up
    to 256 columns, one row per step with a predicated tail up to 8192, two queries per pass beyond 4096); rows that are not whole
    units (100: tests/test_hip_odd_rows.py has the sweep) and the 16-byte generic kernel beyond 8192 columns (8200)."""
    cb, Eb, dewi32, ent32 = _setup(n, dim, seed=dim + n)
    Q = orc.synth_queries(5, dim, seed=dim)
    Qp = device_prepared_queries(Q)
    for k, eta, pref in ((10, 0.3, 0.0), (1, 0.5, 0.0), (100, 0.25, 0.3), (150, 0.5, 0.0)):
        ids, sc = cb.search(Q, k, eta, pref)
        check_batch(Eb, Qp, dewi32, ent32, k, eta, pref, "cosine", ids, sc,
                    min_decisive_frac=0.8 if k <= 10 else 0.5, **TOL)


@pytest.mark.parametrize("dim", [64, 200, 384, 768, 1000, 3072, 5120])
def test_bf16_l2_vs_oracle(dim):
    """space="l2" over a bf16 corpus on the row kernels (one query and groups of 4 / 2 per pass): negated squared distance of
    the bf16-rounded query to the bf16 rows, fp32 sums."""
    from dewi import _engine as eng
    n = 2000 if dim <= 1024 else 700
    rs = np.random.RandomState(dim + 6)
    raw = (rs.randn(n, dim) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    cb = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], "l2").to_bf16()
    Eb = cb.emb.float().cpu().numpy()
    assert np.array_equal(Eb, orc.bf16_round(raw))
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = (rs.randn(5, dim) * 0.5).astype(np.float32)
    Qp = device_prepared_queries(Q, "l2")
    # (scores of hundreds: the gap scales with them, so fewer queries are decisive at large k and wide rows — floors from
    # the oracle alone: 5/5 at k = 10; 4-5 at k = 40; 1-4 at k = 150 up to 3072 columns, 0 at 5120)
    for k in ((10, 40, 150) if dim <= 3072 else (10, 40)):
        ids, sc = cb.search(Q, k, 0.3, 0.0)
        check_batch(Eb, Qp, dewi32, ent32, k, 0.3, 0.0, "l2", ids, sc, min_decisive_frac={10: 0.8, 40: 0.6, 150: 0.2}[k], **TOL)


def test_bf16_sharded_equals_whole():
    from dewi import _engine as eng
    import torch
    n, d, k = 6000, 768, 10
    raw = orc.synth_corpus(n, d, seed=5)
    cols = orc.synth_payload_columns(n, seed=5)
    Q = orc.synth_queries(4, d, seed=6)
    whole = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"]).to_bf16()
    ids_ref, sc_ref = whole.search(Q, k, 0.3, 0.0)
    qd = torch.from_numpy(Q).cuda()
    lists = []
    for lo, hi in ((0, 2500), (2500, 6000)):                       # even boundaries: row pairs stay aligned
        sub = {key: v[lo:hi] for key, v in cols.items()}
        sh = eng.DeviceCorpus.from_host(raw[lo:hi], sub["dewi"], sub["ht_mean"], sub["hi_mean"], id_offset=lo).to_bf16()
        lists.append(sh.candidates_device(qd, 2 * k))
    ids, sc = eng.merge_rerank_device(torch.stack(lists), 2 * k, k, 0.3, 0.0)
    assert np.array_equal(ids.cpu().numpy(), ids_ref) and np.array_equal(sc.cpu().numpy(), sc_ref)


def _random_bf16_cases(n_cases, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_cases):
        n = int(rs.choice([1, 2, 3, 5, 64, 255, 257, 1001, 3000]))
        dim = int(rs.choice([8, 9, 40, 100, 128, 256, 300, 512, 768]))   # >= 8: in 1-3 dimensions bf16 rows collide
                                                                           # exactly and the candidate cut is all ties
        b = int(rs.randint(1, 10))
        k = int(rs.randint(1, min(n, 120) + 1))
        out.append((n, dim, b, k, float(rs.choice([0.0, 0.3, 1.0])), float(rs.choice([0.0, 0.4, -1.0]))))
    return out


@pytest.mark.parametrize("case", _random_bf16_cases(24, seed=4102026), ids=lambda c: "n%d-d%d-b%d-k%d" % c[:4])
def test_bf16_randomised_shapes_vs_oracle(case):
    """Seeded sweep over tiny / odd corpora and dimensions for the bf16 small-batch kernels (pair kernel with and
    without the unpaired last row, 16-byte generic, scalar generic)."""
    n, dim, b, k, eta, pref = case
    cb, Eb, dewi32, ent32 = _setup(n, dim, seed=n * 7 + dim)
    Q = orc.synth_queries(b, dim, seed=dim + b)
    Qp = device_prepared_queries(Q)
    ids, sc = cb.search(Q, k, eta, pref)
    assert ids.shape == (b, k) and ids.min() >= 0 and ids.max() < n
    check_batch(Eb, Qp, dewi32, ent32, k, eta, pref, "cosine", ids, sc, min_decisive_frac=0.7, **TOL)
