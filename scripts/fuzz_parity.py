#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: many seeded random cases of the hot path against the CPU oracle, for a time
budget.  Not part of the test suite (its cases are drawn at run time); a failure prints the case so that it can be
added to tests/ as a fixed one.

    python3 scripts/fuzz_parity.py [--seconds 300] [--seed 1] > gpurun_out/fuzz.log

Case families (all through the C ABI, checked with tests/parity.py's rules or bit for bit where that is the contract):
  search   one corpus, random (rows, dim, element type, space, batch, k, eta, entropy_pref): ids / scores vs the oracle
  shards   the same corpus cut into 2-8 ragged doc-id shards: candidates per shard + merge == the whole-corpus search
  fit      robust fit + score of random signal tables (ties, constants, NaN, tiny n) vs NumPy: medians / MADs bit-exact
"""
import argparse
import sys
import time
import traceback
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd", REPO / "oracle", REPO / "tests"):
    sys.path.insert(0, str(p))

import torch  # noqa: E402

import dewi_oracle as orc  # noqa: E402
from dewi import _engine as eng  # noqa: E402
from dewi.scorer import DewiScorer  # noqa: E402
from parity import compare_query, device_prepared_queries  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="corpora of 64K-160K rows at matrix-core dims, batches up to 300: the batched passes")
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
# (round 4: every kind of row kernel — lanes sharing a short row, one row per step with a predicated tail at every
# instantiated units-per-lane count for fp32 and bf16, the scalar-capable generic kernel, the tuned dim = 256 U ones)
# (and rows that are not whole 16-byte units: fp32 dim % 4 != 0 / bf16 dim % 8 != 0 — 1, 3, 17, 50, 101, 129, 301, 770, 1001, 2050, 3001;
# bf16 also 12, 36, 100, 132, 252, 260, 300)
DIMS = [1, 3, 4, 8, 12, 17, 36, 50, 64, 100, 101, 128, 129, 132, 200, 252, 256, 260, 300, 301, 384, 512, 640, 768, 770, 1000, 1001,
        1024, 1280, 1536, 1792, 2048, 2050, 2304, 3001, 3072, 4096]
fails, done = [], {"search": 0, "shards": 0, "fit": 0}
LAST = {}


def one_search_case(i):
    if args.big:
        # (round 4: widths with a partial last chunk on the depth-split pass — 96, 160, 320, 384, 640, 896, 992)
        # (... and rows that end inside a wave's 32-column slice: 100, 200, 300, 1000, 1004, 36)
        dim = int(rs.choice([96, 128, 160, 256, 256, 320, 384, 384, 512, 512, 640, 768, 768, 768, 896, 992, 1024, 1280, 1312, 1536, 2048,
                             100, 200, 300, 1000, 1004, 36]))
        n = int(rs.randint(65_536, 160_000))
        bf16 = bool(rs.rand() < 0.45)
        space = "l2" if rs.rand() < 0.2 else "cosine"
        b = int(rs.choice([1, 1, 2, 5, 8, 31, 32, 33, 64, 65, 100, 256, 257, 300]))
        k = int(rs.choice([1, 5, 10, 50, 100, 128, 129, 300, 1000]))
    else:
        dim = int(rs.choice(DIMS))
        n = int(np.exp(rs.uniform(0, np.log(min(200_000, 40_000_000 // dim)))))
        bf16 = bool(rs.rand() < 0.4)
        space = "l2" if rs.rand() < 0.3 else "cosine"
        b = int(rs.choice([1, 1, 2, 3, 4, 5, 8, 9, 31, 33, 40, 70]))
        k = int(min(n, np.exp(rs.uniform(0, np.log(3000)))))
    eta = float(rs.choice([0.0, 0.25, 0.3, 0.7, 1.0]))
    pref = float(rs.choice([0.0, 0.0, 0.2, -0.5]))
    case = dict(family="search", i=i, n=n, dim=dim, bf16=bf16, space=space, b=b, k=k, eta=eta, pref=pref)
    LAST.clear()
    LAST.update(case)
    raw = orc.synth_corpus(n, dim, seed=10_000 + i)
    if space == "l2":
        raw = raw * rs.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    Q = orc.synth_queries(b, dim, seed=20_000 + i)
    cols = orc.synth_payload_columns(n, seed=10_000 + i)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], space=space)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    kw = {}
    if bf16:
        c = c.to_bf16()
        E, Qo = c.emb.float().cpu().numpy(), device_prepared_queries(Q, space)
        kw = dict(gap=1e-6, score_tol=1e-5, prepared=True)
    else:
        E, Qo = c.emb.cpu().numpy(), Q
        if space == "cosine" and dim % 8 == 0 and 136 <= dim <= 1536 and rs.rand() < 0.7:
            c.enable_bf16_shadow(single_query=True)   # matrix-core pass over the bf16 shadow + exact re-scoring (1+ queries)
            LAST.update(shadow=True)
            done["shadow"] = done.get("shadow", 0) + 1
    ids, sc = c.search(Q, k, eta, pref)
    assert ids.shape == (b, k)
    check = range(b) if b <= 16 else sorted(rs.choice(b, 16, replace=False).tolist())   # the oracle costs n*dim per query
    for j in check:
        _, msg = compare_query(E, Qo[j], dewi32, ent32, k, eta, pref, space, ids[j], sc[j], exact_gaps=n * dim <= 4_000_000, **kw)
        assert msg is None, f"query {j}: {msg}"
    done["search"] += 1
    # the same corpus as ragged shards: candidates + merge must equal the whole-corpus answer bit for bit
    if n >= 8 and rs.rand() < 0.6:
        case["family"] = "shards"
        s = int(rs.randint(2, 9))
        cuts = sorted(set([0, n] + rs.randint(1, n, size=s - 1).tolist()))
        cc = min(2 * k, n)
        qd = torch.from_numpy(Q).cuda()
        lists, families = [], set()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sh = eng.DeviceCorpus(c.emb[lo:hi], c.dewi32[lo:hi], c.ent32[lo:hi], space, id_offset=lo)   # (no shadow: shard records)
            lists.append(sh.candidates_device(qd, cc))
            families.add(sh.scan_kernel_name(b, k, candidates=cc).split("<")[0].startswith("mfma"))
        case["cuts"] = cuts
        LAST.update(family="shards", cuts=cuts)
        m_ids, m_sc = eng.merge_rerank_device(torch.stack(lists), cc, k, eta, pref)
        w_ids, w_sc = c.search_device(qd, k, eta, pref)
        torch.cuda.synchronize()
        ok = w_ids[:, 0] >= 0                                   # (a refused matrix-core query has no answer to compare)
        m_ok = m_ids[:, 0] >= 0
        both = (ok & m_ok).cpu().numpy()
        # per-row sums do not depend on the shard a row is in when the same kernel family serves both; a shard too
        # small for the matrix-core pass takes the row kernels (other summation order): compare through the oracle then
        # (which family serves a shape is the library's own answer, dewi_knn_scan_kernel: since round 4 the batch size from which
        # the matrix-core pass is taken depends on the element type, the space and the width)
        families.add(c.scan_kernel_name(b, k).split("<")[0].startswith("mfma"))
        same_path = len(families) == 1 and c.shadow is None
        # the bf16 row kernel takes 1536-byte rows in pairs: a row's sum depends on its place in the pair, so only
        # shards that start on even rows (what dewi.sharded.shard_bounds produces) are bit-equal to the whole there
        pairs_kept = not (bf16 and dim == 768) or all(lo % 2 == 0 for lo in cuts[:-1])
        if same_path and pairs_kept:
            assert torch.equal(m_ids[both], w_ids[both]) and torch.equal(m_sc[both], w_sc[both]), "shards != whole"
        else:
            mi, ms = m_ids.cpu().numpy(), m_sc.cpu().numpy()
            for j in [j for j in np.nonzero(both)[0] if j in set(check)]:
                _, msg = compare_query(E, Qo[j], dewi32, ent32, k, eta, pref, space, mi[j], ms[j], exact_gaps=False, **kw)
                assert msg is None, f"sharded query {j}: {msg}"
        done["shards"] += 1
    return case


def one_fit_case(i):
    n = int(np.exp(rs.uniform(0, np.log(300_000))))
    case = dict(family="fit", i=i, n=n)
    LAST.clear()
    LAST.update(case)
    cols = {}
    for key in orc.SIGNAL_KEYS:
        kind = rs.randint(6)
        if kind == 0:
            v = rs.gamma(2, 0.5, n)
        elif kind == 1:
            v = np.round(rs.rand(n) * 4) / 4              # heavy ties
        elif kind == 2:
            v = np.full(n, 0.37)                           # constant: MAD 0 -> 1e-8
        elif kind == 3:
            v = np.sort(rs.randn(n))                       # sorted
        elif kind == 4:
            v = rs.randn(n) * 10.0 ** rs.randint(-6, 6)
        else:
            v = rs.beta(1, 10, n)
        cols[key] = v.astype(np.float32)
    if rs.rand() < 0.1 and n > 2:
        cols[orc.SIGNAL_KEYS[int(rs.randint(7))]][int(rs.randint(n))] = np.nan
    case["kinds"] = "mixed"
    sc = DewiScorer()
    dev = rs.rand() < 0.5
    table = torch.from_numpy(np.stack([cols[key] for key in orc.SIGNAL_KEYS])).cuda()
    dcols = {key: table[j] for j, key in enumerate(orc.SIGNAL_KEYS)}
    sc.fit_stats_columns(dcols if dev else cols)
    with np.errstate(invalid="ignore"):
        med, mad = orc.robust_fit(cols)
    for key in orc.SIGNAL_KEYS:
        a, bb = sc.stats.medians[key], med[key]
        assert (a == bb) or (np.isnan(a) and np.isnan(bb)), (key, a, bb)
        a, bb = sc.stats.mads[key], mad[key]
        assert (a == bb) or (np.isnan(a) and np.isnan(bb)), (key, a, bb)
    got = sc.score_batch(dcols if dev else cols)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        ref = orc.score({key: v.astype(np.float64) for key, v in cols.items()}, med, mad)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    if fin.any():
        assert np.max(np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])) < 1e-15
    done["fit"] += 1
    return case


t_end = time.time() + args.seconds
i = 0
while time.time() < t_end:
    i += 1
    case = None
    try:
        case = one_fit_case(i) if i % 4 == 0 else one_search_case(i)
    except Exception as e:  # noqa: BLE001
        fails.append((i, dict(LAST), repr(e)[:600]))
        print(f"FAIL case {i} {LAST}: {repr(e)[:600]}", flush=True)
        traceback.print_exc(limit=2)
    if i % 20 == 0:
        print(f"[{time.time() - (t_end - args.seconds):6.0f} s] {i} cases, {done}, {len(fails)} failures", flush=True)
print(f"DONE: {i} cases, {done}, {len(fails)} failures")
for f in fails:
    print("  ", f)
sys.exit(1 if fails else 0)
