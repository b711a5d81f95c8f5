"""GPU parity tests of the scorer path (A6-A8) through the C ABI, against the reference's
own outputs (tests/golden/g4_scorer.*) and the CPU oracle."""
import json

import numpy as np
import pytest

import dewi_oracle as orc

pytestmark = pytest.mark.gpu


def _weights(w):
    from dewi.types import Weights
    return None if w is None else Weights(**w)


def test_g4_fit_and_score_match_reference(golden, golden_dir):
    from dewi.scorer import DewiScorer
    g = golden("g4_scorer.npz")
    meta = json.loads((golden_dir / "g4_scorer.json").read_text())
    tags = [t for t in meta if isinstance(meta[t], dict) and "keys" in meta[t]]
    for tag in tags:
        m = meta[tag]
        keys = m["keys"]
        cols = {k: g[f"{tag}__in_{k}"] for k in keys}
        s = DewiScorer(weights=_weights(m["weights"]), delta=m["delta"])
        s.fit_stats_columns(cols)
        med = np.array([s.stats.medians[k] for k in keys])
        mad = np.array([s.stats.mads[k] for k in keys])
        assert np.array_equal(med, g[f"{tag}__med"]), tag          # exact fp32 order statistics
        assert np.array_equal(mad, g[f"{tag}__mad"]), tag
        sig = {k: cols[k] for k in orc.SIGNAL_KEYS}
        for mode, key in (("standard", "score"), ("conditional", "cond")):
            got = s.score_batch(sig, mode)
            ref = g[f"{tag}__{key}"]
            assert got.dtype == np.float64 and got.shape == ref.shape
            err = np.max(np.abs(got - ref) / np.abs(ref))
            assert err <= 4e-16, (tag, mode, err)                    # <= 2 ulp of float64 (exp)
        # fp32 columns take the fp32 device path and must give the same numbers (inputs are fp32-representable)
        sig32 = {k: cols[k].astype(np.float32) for k in orc.SIGNAL_KEYS}
        assert np.array_equal(s.score_batch(sig32, "standard"), s.score_batch(sig, "standard"))


def test_row_api_and_reference_quirks(golden_dir):
    """tests/test_scorer_weights.py of the reference, with the values it actually produces."""
    from dewi.scorer import DewiScorer
    from dewi.types import Payload, Weights
    meta = json.loads((golden_dir / "g4_scorer.json").read_text())
    w = Weights(alpha_t=0.6, alpha_i=0.2, alpha_r=0.2, alpha_n=0.1)
    s = DewiScorer(weights=w)
    p = Payload(ht_mean=1.0, hi_mean=0.5, redundancy=0.2, noise=0.1, ht_q90=1.2, hi_q90=0.7)
    sig = p.to_dict()
    sig["I_hat"] = 0.0
    with pytest.raises(AssertionError):
        s.score(sig)                                         # score before fit (scorer.py:50)
    s.fit_stats([sig])
    v, c = s.score(sig), s.score_conditional(sig)
    assert isinstance(v, float) and isinstance(c, float)
    lit = meta["literal_row"]
    assert s.stats.medians == lit["medians"] and s.stats.mads == lit["mads"]
    assert v == pytest.approx(lit["score"], rel=1e-15) and c == pytest.approx(lit["cond"], rel=1e-15)
    assert v == pytest.approx(0.30275610348537835, rel=1e-15)   # fp32 fit vs f64 z: not 0.5
    # an fp32-representable row gives exactly 0.5 (MAD = 0 -> 1e-8, z = 0)
    sig2 = {k: float(np.float32(x)) for k, x in sig.items()}
    s.fit_stats([sig2])
    assert s.score(sig2) == 0.5
    with pytest.raises(KeyError):
        s.score({"ht_mean": 1.0})
    # ctor overwrites weights.delta and mutates the caller's object (scorer.py:37-40)
    w5 = Weights(delta=5.0)
    assert DewiScorer(weights=w5).weights.delta == 3.0 and w5.delta == 3.0
    # even-N median: fp32 mean of the two middles
    from dewi.scorer import RobustStats
    ev = meta["even_median"]
    st = RobustStats.fit([{"x": x} for x in ev["values"]])
    assert st.medians["x"] == ev["med"] and st.mads["x"] == ev["mad"]
    assert st.z("x", 0.5) == (0.5 - ev["med"]) / (1.4826 * ev["mad"])


@pytest.mark.parametrize("n", [1, 2, 3, 64, 1000, 65537, 1_000_000])
def test_fit_sizes_and_special_values_vs_numpy(n):
    """Exact medians at ragged sizes up to C5's 1M, with duplicates, negatives, zeros and infinities."""
    from dewi.scorer import RobustStats
    rs = np.random.RandomState(n)
    cols = {
        "gauss": rs.randn(n).astype(np.float32),
        "dups": rs.randint(0, 5, n).astype(np.float32),
        "gamma": rs.gamma(2, 0.5, n).astype(np.float32),
        "signed_zero": np.where(rs.rand(n) < 0.5, -0.0, 0.0).astype(np.float32),
        "with_inf": np.where(rs.rand(n) < 0.01, np.inf, rs.randn(n)).astype(np.float32),
        "tiny": (rs.randn(n) * 1e-40).astype(np.float32),       # subnormals
    }
    st = RobustStats.fit_columns(cols)
    med, mad = orc.robust_fit(cols)
    for k in cols:
        assert st.medians[k] == med[k], (k, st.medians[k], med[k])
        assert st.mads[k] == mad[k] or (np.isnan(st.mads[k]) and np.isnan(mad[k])), (k, st.mads[k], mad[k])


def test_fit_nan_column_gives_nan():
    from dewi.scorer import RobustStats
    x = np.arange(100, dtype=np.float32)
    x[3] = np.nan
    st = RobustStats.fit_columns({"a": x, "b": np.arange(100, dtype=np.float32)})
    assert np.isnan(st.medians["a"]) and np.isnan(st.mads["a"])
    assert st.medians["b"] == 49.5 and st.mads["b"] == 25.0


def test_score_1m_rows_matches_oracle():
    """Config C5's scorer part: N = 1M, 7 signals, fit + score on the GPU vs NumPy."""
    from dewi.scorer import DewiScorer
    n = 1_000_000
    cols64 = orc.synth_payload_columns(n, seed=5)
    cols = {k: cols64[k].astype(np.float32) for k in orc.SIGNAL_KEYS}
    s = DewiScorer()
    s.fit_stats_columns(cols)
    med, mad = orc.robust_fit(cols)
    assert s.stats.medians == med and s.stats.mads == mad
    for mode in ("standard", "conditional"):
        got = s.score_batch(cols, mode)
        ref = orc.score({k: v.astype(np.float64) for k, v in cols.items()}, med, mad, None, 3.0, mode)
        assert np.max(np.abs(got - ref) / ref) <= 4e-16


@pytest.mark.parametrize("n", [65_537, 100_003, 400_000, 1_000_000])
def test_fast_fit_adversarial_columns_stay_exact(n):
    """The two-launch fit (csrc/robust_fit_fast.hip) guesses a bracket from a strided sample and VERIFIES it; when
    the guess is wrong — data arranged against the sample — or a buffer overflows, the last workgroup selects over
    the whole column instead.  Either way the order statistics must be NumPy's, bit for bit."""
    from dewi.scorer import RobustStats
    rs = np.random.RandomState(n)
    stride = n // 8192
    periodic = rs.randn(n).astype(np.float32)
    periodic[(np.arange(8192) * 2 + 1) * n // (2 * 8192)] = 1e6           # every SAMPLED position holds an outlier
    runs = rs.randn(n).astype(np.float32)                                 # round 4: the sample is 256 runs of 16 values
    run_start = (((2 * np.arange(256) + 1) * n) >> 9) & ~15
    runs[np.minimum((run_start[:, None] + np.arange(16)[None, :]).ravel(), n - 1)] = -1e6   # every sampled value an outlier
    heavy_zero = np.where(rs.rand(n) < 0.7, 0.0, rs.gamma(2, 0.5, n)).astype(np.float32)   # 70 % ties at the median
    few_values = rs.choice(np.array([0.25, 0.5, 0.75], np.float32), n)     # three distinct values: lo == hi likely
    bimodal = np.where(np.arange(n) % 2 == 0, rs.randn(n) - 50, rs.randn(n) + 50).astype(np.float32)   # empty middle
    cols = {
        "sorted_up": np.sort(rs.randn(n).astype(np.float32)),
        "sorted_down": np.sort(rs.randn(n).astype(np.float32))[::-1].copy(),
        "periodic_outliers": periodic,
        "outliers_on_the_sampled_runs": runs,
        "slow_drift": (np.linspace(-3, 3, n) + 0.01 * rs.randn(n)).astype(np.float32),   # runs of near-equal values
        "heavy_zero": heavy_zero,
        "few_values": few_values,
        "constant": np.full(n, 3.25, np.float32),
        "bimodal": bimodal,
        "blocks": np.repeat(rs.randn(n // stride + 1).astype(np.float32), stride)[:n].copy(),   # runs of equal values
        "negatives": (-rs.gamma(2, 0.5, n)).astype(np.float32),
        "mixed_nan": np.where(np.arange(n) == n // 3, np.nan, rs.randn(n)).astype(np.float32),
    }
    st = RobustStats.fit_columns(cols)
    med, mad = orc.robust_fit(cols)
    for k in cols:
        same_med = st.medians[k] == med[k] or (np.isnan(st.medians[k]) and np.isnan(med[k]))
        same_mad = st.mads[k] == mad[k] or (np.isnan(st.mads[k]) and np.isnan(mad[k]))
        assert same_med and same_mad, (k, st.medians[k], med[k], st.mads[k], mad[k])


def test_fast_fit_through_the_c_abi_with_a_leading_dimension():
    """dewi_robust_fit_f32 with ld > n (columns not 16-byte aligned): head / body / tail of every slice exactly once."""
    import torch
    from dewi import _native as nat
    lib = nat.load_library()
    n, ld, ns = 300_001, 300_007, 3
    rs = np.random.RandomState(1)
    host = np.zeros((ns, ld), np.float32)
    host[:, :n] = rs.gamma(2, 0.5, (ns, n)).astype(np.float32)
    host[:, n:] = 1e9                                   # padding behind the columns must not be read
    dev = torch.from_numpy(host).cuda()
    med = torch.empty(ns, dtype=torch.float32, device="cuda")
    mad = torch.empty(ns, dtype=torch.float32, device="cuda")
    wsb = int(lib.dewi_robust_fit_workspace_bytes(ns))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    for _ in range(3):                                  # repeated calls reuse the workspace (counters are re-zeroed)
        nat.check(lib.dewi_robust_fit_f32(nat.ptr(dev), n, ld, ns, nat.ptr(med), nat.ptr(mad), nat.ptr(ws), wsb, nat.stream_ptr()))
    m, d = med.cpu().numpy(), mad.cpu().numpy()
    for s in range(ns):
        want = np.median(host[s, :n])
        assert m[s] == want and d[s] == np.median(np.abs(host[s, :n] - want))
