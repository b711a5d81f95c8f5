"""GPU parity tests of the row kernels for rows that are NOT whole 16-byte units (fp32: dim % 4 != 0, bf16: dim % 8 != 0),
through the C ABI.

The reference scores any width with one BLAS call (src/dewi/backends.py:431-436).  Here such rows take the PH = true forms of
the any-width kernels (csrc/scan_any.hpp): aligned 16-byte loads, every wave / lane group on the rows of one residue mod
G = 16 / gcd(16, row bytes), the query fragments shifted by that residue's offset, the neighbouring rows' columns in a row's
first and last unit masked off (narrow rows with one query: a wave on consecutive rows with a register set of fragments per
residue, scan_rows_odd_contig).  What can go wrong there and nowhere else: a wrong shift for one residue, a neighbour's
columns leaking into a sum (NaN / huge neighbours), the first and the last row of the buffer, shards whose first row does not
start on a unit, fewer rows than residues.
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import check_batch, compare_query, device_prepared_queries

pytestmark = pytest.mark.gpu
BF16_TOL = dict(gap=1e-6, score_tol=1e-5, prepared=True)

# fp32: every residue of dim mod 4, short rows (lanes sharing a row: 1 .. 32 units), one row per wave step at every
# units-per-lane count that is instantiated (1 .. 16), a row that just fits 1024 units (4090), one that does not (4101: generic)
F32_DIMS = [5, 6, 7, 9, 10, 17, 30, 50, 63, 65, 101, 126, 127, 129, 130, 255, 257, 301, 515, 770, 1001, 1030, 1283, 1541, 2049,
            2050, 2571, 3001, 4090, 4101]
# bf16: every residue of dim mod 8 among them
BF16_DIMS = [4, 9, 12, 20, 30, 63, 100, 129, 250, 300, 301, 515, 772, 1001, 1030, 2052, 3003, 5001, 8180, 8201]


def _eng():
    from dewi import _engine
    return _engine


def _kernel(c, b, k):
    return c.scan_kernel_name(b, k)


def _expect_odd_kernel(c, dim, limit):
    name = _kernel(c, 1, 10)
    if dim <= limit:
        # the PH = true form (or, narrow rows with one query, its consecutive-rows variant), not scan_generic_*
        assert (name.endswith("true>") and "any<" in name) or name.startswith("scan_rows_odd_contig<"), name
    else:
        assert name.startswith("scan_generic"), name


@pytest.mark.parametrize("dim", F32_DIMS)
def test_odd_widths_fp32_cosine_vs_oracle(dim):
    n = 3001 if dim <= 1100 else 1200
    raw = orc.synth_corpus(n, dim, seed=dim)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = orc.synth_queries(5, dim, seed=dim + 1)      # 5 = one four-query pass + a one-query pass where four fit
    c = _eng().DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    _expect_odd_kernel(c, dim, 4090)
    E = c.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for k, eta, pref in ((1, 0.3, 0.0), (10, 0.3, 0.0), (10, 0.7, -0.5), (100, 0.25, 0.3), (150, 0.5, 0.0)):
        ids, sc = c.search(Q, k, eta, pref)          # c = 2, 20 (one list per workgroup), 200 (per wave), 300 (dense keys)
        check_batch(E, Q, dewi32, ent32, k, eta, pref, "cosine", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.6)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_tiny_odd_widths_fp32(dim):
    """Rows shorter than a unit: up to four rows inside one 16-byte load, both masks in the same unit.  (Cosine similarities of
    such rows are all +-1 or close: nothing decisive to compare id for id, the harness compares scores and tied id sets.)"""
    n = 1003
    rs = np.random.RandomState(dim)
    raw = rs.randn(n, dim).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = rs.randn(5, dim).astype(np.float32)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for space in ("cosine", "l2"):
        c = _eng().DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space)
        assert _kernel(c, 1, 10).endswith("true>")       # (rows shorter than a unit: lanes sharing a row)
        E = c.emb.cpu().numpy()
        for k in (1, 10, 150):
            ids, sc = c.search(Q, k, 0.3, 0.0)
            for j in range(5):
                _, msg = compare_query(E, Q[j], dewi32, ent32, k, 0.3, 0.0, space, ids[j], sc[j])
                assert msg is None, (space, k, j, msg)


@pytest.mark.parametrize("dim", [5, 30, 101, 129, 301, 1001, 2050, 3001])
def test_odd_widths_fp32_l2_vs_oracle(dim):
    n = 2000 if dim <= 1100 else 1000
    rs = np.random.RandomState(dim)
    raw = (rs.randn(n, dim) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=3)
    Q = (rs.randn(5, dim) * 0.5).astype(np.float32)
    c = _eng().DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], "l2")
    _expect_odd_kernel(c, dim, 4090)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for k in (10, 40, 150):
        ids, sc = c.search(Q, k, 0.3, 0.0)
        check_batch(raw, Q, dewi32, ent32, k, 0.3, 0.0, "l2", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.4)


@pytest.mark.parametrize("dim", BF16_DIMS)
def test_odd_widths_bf16_vs_oracle(dim):
    eng = _eng()
    n = 2001 if dim <= 1100 else 700
    raw = orc.synth_corpus(n, dim, seed=dim + 11)
    cols = orc.synth_payload_columns(n, seed=dim)
    cb = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"]).to_bf16()
    _expect_odd_kernel(cb, dim, 8180)
    Eb = cb.emb.float().cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = orc.synth_queries(5, dim, seed=dim)
    Qp = device_prepared_queries(Q)
    for k, eta, pref in ((10, 0.3, 0.0), (1, 0.5, 0.0), (100, 0.25, 0.3), (150, 0.5, 0.0)):
        ids, sc = cb.search(Q, k, eta, pref)
        check_batch(Eb, Qp, dewi32, ent32, k, eta, pref, "cosine", ids, sc, min_decisive_frac=0.8 if k <= 10 else 0.5, **BF16_TOL)


@pytest.mark.parametrize("dim", [12, 100, 301, 1030, 3003])
def test_odd_widths_bf16_l2_vs_oracle(dim):
    """bf16 l2: the four-query pass is four one-query launches here (registers), same answers."""
    eng = _eng()
    n = 2000 if dim <= 1100 else 700
    rs = np.random.RandomState(dim + 6)
    raw = (rs.randn(n, dim) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    cb = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], "l2").to_bf16()
    Eb = cb.emb.float().cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    Q = (rs.randn(5, dim) * 0.5).astype(np.float32)
    Qp = device_prepared_queries(Q, "l2")
    for k in (10, 40, 150):
        ids, sc = cb.search(Q, k, 0.3, 0.0)
        check_batch(Eb, Qp, dewi32, ent32, k, 0.3, 0.0, "l2", ids, sc, min_decisive_frac={10: 0.8, 40: 0.6, 150: 0.2}[k], **BF16_TOL)


@pytest.mark.parametrize("dim,bf16", [(7, False), (30, False), (101, False), (301, False), (1001, False), (2050, False),
                                       (12, True), (100, True), (301, True), (1001, True)])
def test_neighbour_rows_do_not_leak(dim, bf16):
    """A row's first and last 16-byte unit also hold columns of the rows before and after it.  A zero row is NaN after the
    build's normalisation (reference: no guard, backends.py:403-407), and in l2 a row of 1e30 squares to +inf: neither may
    change the score of the row next to it — every other row must score exactly as in a corpus without the poison."""
    eng = _eng()
    n = 777
    rs = np.random.RandomState(dim)
    raw = rs.randn(n, dim).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = rs.randn(3, dim).astype(np.float32)
    poison = [0, 13, 14, 400, 401, 402, 403, 404, 405, 406, 407, n - 1]   # first row, last row, a pair, a run of all residues
    for space, value in (("cosine", 0.0), ("l2", 1e30)):
        bad = raw.copy()
        bad[poison] = value
        clean = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space)
        dirty = eng.DeviceCorpus.from_host(bad, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space)
        if bf16:
            clean, dirty = clean.to_bf16(), dirty.to_bf16()
        assert _kernel(dirty, 1, 10).endswith("true>") or _kernel(dirty, 1, 10).startswith("scan_rows_odd_contig<")
        k = n                                              # every row's score comes back (eta = 0: score == similarity)
        ids_c, sc_c = clean.search(Q, k, 0.0, 0.0)
        ids_d, sc_d = dirty.search(Q, k, 0.0, 0.0)
        for j in range(3):
            by_id_c = dict(zip(ids_c[j].tolist(), sc_c[j].tolist()))
            by_id_d = dict(zip(ids_d[j].tolist(), sc_d[j].tolist()))
            assert len(by_id_d) == n
            for r in range(n):
                if r in poison:
                    assert np.isnan(by_id_d[r]) if space == "cosine" else by_id_d[r] == -np.inf, (space, r, by_id_d[r])
                else:
                    assert by_id_d[r] == by_id_c[r], (space, j, r, by_id_d[r], by_id_c[r])


@pytest.mark.parametrize("dim,bf16", [(129, False), (255, False), (301, False), (387, False), (300, True), (250, True)])
def test_consecutive_rows_form_equals_the_per_residue_form(dim, bf16):
    """Narrow odd rows: ONE query runs with a wave on consecutive rows (scan_rows_odd_contig, a register set of query fragments
    per residue), four queries per pass with a wave per residue (scan_rows_any<..., true>).  Same loads, same lanes, same
    order of additions per row: the two forms must agree bit for bit — ids and scores, every list shape, both spaces."""
    eng = _eng()
    n = 5003
    rs = np.random.RandomState(dim)
    raw = rs.randn(n, dim).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = rs.randn(4, dim).astype(np.float32)
    for space in ("cosine", "l2"):
        c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space)
        if bf16:
            c = c.to_bf16()
        assert _kernel(c, 1, 10).startswith("scan_rows_odd_contig<")
        four_ways = bf16 and space == "l2"                # (bf16 l2 serves four queries as four one-query launches)
        assert four_ways or ("any<" in _kernel(c, 4, 10) and ", 4, " in _kernel(c, 4, 10)), _kernel(c, 4, 10)
        for k in (10, 100, 150):                             # one list per workgroup, per wave, dense keys
            ids4, sc4 = c.search(Q, k, 0.3, 0.1)
            for j in range(4):
                ids1, sc1 = c.search(Q[j], k, 0.3, 0.1)
                assert np.array_equal(ids1[0], ids4[j]) and np.array_equal(sc1[0], sc4[j]), (space, k, j)


@pytest.mark.parametrize("dim,bf16", [(5, False), (30, False), (101, False), (129, False), (301, False), (1001, False), (9, True), (100, True),
                                       (301, True)])
def test_odd_width_shards_equal_the_whole(dim, bf16):
    """Shards are views into the same buffer: their first row starts anywhere inside a unit.  A row's offset inside its first
    unit is a property of its ADDRESS, so the same lanes sum the same columns in the shard and in the whole: bit-equal."""
    import torch
    eng = _eng()
    n, k, b = 6007, 10, 5
    raw = orc.synth_corpus(n, dim, seed=dim + 5)
    cols = orc.synth_payload_columns(n, seed=dim)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    if bf16:
        c = c.to_bf16()
    Q = orc.synth_queries(b, dim, seed=dim)
    qd = torch.from_numpy(Q).cuda()
    w_ids, w_sc = c.search_device(qd, k, 0.3, 0.2)
    cc = 2 * k
    for cuts in ([0, 1, 2, 3, 4, 5, 6, 7, 8, n], [0, 2999, 3000, 3011, n], [0, 1001, 1003, 4006, 4009, n]):
        lists = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sh = eng.DeviceCorpus(c.emb[lo:hi], c.dewi32[lo:hi], c.ent32[lo:hi], "cosine", id_offset=lo)
            assert sh.emb.data_ptr() == c.emb.data_ptr() + lo * dim * c.emb.element_size()
            lists.append(sh.candidates_device(qd, cc))
        m_ids, m_sc = eng.merge_rerank_device(torch.stack(lists), cc, k, 0.3, 0.2)
        assert torch.equal(m_ids, w_ids) and torch.equal(m_sc, w_sc), cuts


@pytest.mark.parametrize("n", [1, 2, 3, 7, 9, 65])
def test_fewer_rows_than_residues(n):
    """Odd widths repeat their offsets every 2, 4 (fp32) or 8 (bf16) rows; a corpus with fewer rows leaves residues empty."""
    eng = _eng()
    for dim, bf16 in ((5, False), (101, False), (1001, False), (9, True), (301, True)):
        rs = np.random.RandomState(100 * n + dim)
        raw = rs.randn(n, dim).astype(np.float32)
        cols = orc.synth_payload_columns(n, seed=n)
        Q = rs.randn(2, dim).astype(np.float32)
        c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
        dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
        kw = {}
        if bf16:
            c = c.to_bf16()
            E, Qo, kw = c.emb.float().cpu().numpy(), device_prepared_queries(Q), BF16_TOL
        else:
            E, Qo = c.emb.cpu().numpy(), Q
        k = min(n, 5)
        ids, sc = c.search(Q, k, 0.3, 0.0)
        assert sorted(ids[0].tolist()) == sorted(set(ids[0].tolist())) and ids.max() < n and ids.min() >= 0
        for j in range(2):
            _, msg = compare_query(E, Qo[j], dewi32, ent32, k, 0.3, 0.0, "cosine", ids[j], sc[j], **kw)
            assert msg is None, (n, dim, j, msg)
