#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports ``dewi`` from /root/reference/src (never shipped,
never copied).  The outputs are data — inputs, ids, scores, medians — and are
committed; this script is committed beside them so they can be regenerated:

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py

Vector families:
  g1  tests/test_index.py shape: N=100, d=128, 5 queries, k=10, eta x pref grid
  g2  C1: 10 000 x 768, 64 queries, k=10, eta=0.3 (inputs regenerated from seed)
  g3  edge cases: k==N, k>N raises, zero query, l2, pref != 0, N < 2k
  g4  scorer: fit medians/MADs + score/score_conditional, N in {1,2,101,10000}
  g5  persistence: ExactIndex.save and DewiIndex.save directories
"""
from __future__ import annotations

import json
import os
import shutil
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REF_SRC = Path("/root/reference/src")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF_SRC))
sys.path.insert(0, str(REPO / "oracle"))

import logging  # noqa: E402

logging.disable(logging.WARNING)

from dewi.index import DewiIndex, ExactIndex  # noqa: E402  (the reference)
from dewi.scorer import DewiScorer  # noqa: E402
from dewi.types import Payload, Weights  # noqa: E402

import dewi_oracle as orc  # noqa: E402  (only for the shared synthetic-input generators)

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
PKEYS = orc.PAYLOAD_KEYS


def payloads_from_columns(cols, n):
    return [Payload(**{k: float(cols[k][i]) for k in PKEYS}) for i in range(n)]


def build_ref(E, payloads, space="cosine", ids=None):
    idx = ExactIndex(dim=E.shape[1], space=space)
    ids = ids or [f"doc_{i:08d}" for i in range(E.shape[0])]
    for i, (row, p) in enumerate(zip(E, payloads)):
        idx.add(ids[i], row, p)
    idx.build()
    return idx, {d: i for i, d in enumerate(ids)}


def run(idx, pos, q, k, eta, pref):
    res = idx.search(q, k=k, eta=eta, entropy_pref=pref)
    ids = np.array([pos[r[0]] for r in res], dtype=np.int64)
    sc = np.array([r[1] for r in res], dtype=np.float64)
    assert np.all(sc == sc.astype(np.float32).astype(np.float64)) or np.any(np.isnan(sc))
    return ids, sc.astype(np.float32)


# ------------------------------------------------------------------ g1
def g1():
    rs = np.random.RandomState(42)
    n, d, nq, k = 100, 128, 5, 10
    E = rs.randn(n, d).astype(np.float32)
    E = E / np.linalg.norm(E, axis=1, keepdims=True)
    cols = {key: np.empty(n) for key in PKEYS}
    for i in range(n):  # per-doc draw order of tests/test_index.py:36-50
        cols["dewi"][i] = float(np.clip(rs.beta(2, 2), 0, 1))
        cols["ht_mean"][i] = float(rs.gamma(2, 0.5))
        cols["ht_q90"][i] = float(rs.gamma(2, 0.5) * 1.5)
        cols["hi_mean"][i] = float(rs.gamma(2, 0.3))
        cols["hi_q90"][i] = float(rs.gamma(2, 0.3) * 1.5)
        cols["I_hat"][i] = float(rs.beta(2, 2))
        cols["redundancy"][i] = float(rs.beta(1, 5))
        cols["noise"][i] = float(rs.beta(1, 10))
    Q = rs.randn(nq, d).astype(np.float32)
    Q = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    idx, pos = build_ref(E, payloads_from_columns(cols, n))
    etas = [0.0, 0.3, 0.5, 1.0]
    prefs = [-1.0, 0.0, 0.5, 1.0]
    ids = np.zeros((len(etas), len(prefs), nq, k), np.int64)
    sc = np.zeros((len(etas), len(prefs), nq, k), np.float32)
    for a, eta in enumerate(etas):
        for b, pref in enumerate(prefs):
            for j in range(nq):
                ids[a, b, j], sc[a, b, j] = run(idx, pos, Q[j], k, eta, pref)
    np.savez_compressed(OUT / "g1_test_index_shape.npz", E=E, Q=Q, etas=np.array(etas), prefs=np.array(prefs),
                        k=np.int64(k), ids=ids, scores=sc, stored=np.asarray(idx._embeddings),
                        **{f"p_{key}": cols[key] for key in PKEYS})
    print("g1 ok")


# ------------------------------------------------------------------ g2
def g2():
    n, d, nq, k, eta = 10_000, 768, 64, 10, 0.3
    E = orc.synth_corpus(n, d, seed=42)
    cols = orc.synth_payload_columns(n, seed=42)
    Q = orc.synth_queries(nq, d, seed=7)
    idx, pos = build_ref(E, payloads_from_columns(cols, n))
    ids = np.zeros((nq, k), np.int64)
    sc = np.zeros((nq, k), np.float32)
    cand = np.zeros((nq, 2 * k), np.int64)
    csim = np.zeros((nq, 2 * k), np.float32)
    stored = np.asarray(idx._embeddings)
    for j in range(nq):
        ids[j], sc[j] = run(idx, pos, Q[j], k, eta, 0.0)
        # the candidate set the reference's steps 1-4 produce (backends.py:420-447),
        # re-derived with the reference's own stored matrix and the same NumPy calls
        q = Q[j] / np.linalg.norm(Q[j])
        s = np.dot(stored, q.reshape(1, -1).T).flatten()
        top = np.argpartition(s, -2 * k)[-2 * k:]
        o = np.argsort(-s[top], kind="stable")
        cand[j], csim[j] = top[o], s[top][o]
    # checksum of the regenerated inputs so a drifting generator is caught
    np.savez_compressed(OUT / "g2_c1_10k_768.npz", n=np.int64(n), d=np.int64(d), k=np.int64(k), eta=np.float64(eta),
                        corpus_seed=np.int64(42), query_seed=np.int64(7), ids=ids, scores=sc, cand_ids=cand,
                        cand_sims=csim, e_sum=np.float64(E.astype(np.float64).sum()),
                        q_sum=np.float64(Q.astype(np.float64).sum()),
                        stored_rows_0_3=stored[:4].copy(), dewi_sum=np.float64(cols["dewi"].sum()))
    print("g2 ok")


# ------------------------------------------------------------------ g3
def g3():
    rs = np.random.RandomState(1234)
    n, d = 50, 16
    E = rs.randn(n, d).astype(np.float32) * 3.0  # un-normalised on purpose (l2 uses raw rows)
    cols = orc.synth_payload_columns(n, seed=5)
    pay = payloads_from_columns(cols, n)
    Q = rs.randn(4, d).astype(np.float32)
    cases = {}
    idx, pos = build_ref(E, pay)
    idx_l2, pos_l2 = build_ref(E, pay, space="l2")

    def put(name, index, p, q, k, eta, pref):
        ids, sc = run(index, p, q, k, eta, pref)
        cases[name] = dict(q=q, k=k, eta=eta, pref=pref, ids=ids, scores=sc)

    put("k_eq_n", idx, pos, Q[0], n, 0.3, 0.0)
    put("n_lt_2k", idx, pos, Q[1], 30, 0.3, 0.0)            # c = min(60, 50) = 50
    put("pref_pos", idx, pos, Q[2], 10, 0.3, 0.7)
    put("pref_neg", idx, pos, Q[2], 10, 0.25, -0.4)
    put("eta_one", idx, pos, Q[3], 10, 1.0, 0.0)
    put("eta_zero", idx, pos, Q[3], 10, 0.0, 0.0)
    put("k_one", idx, pos, Q[0], 1, 0.5, 0.0)
    put("l2_basic", idx_l2, pos_l2, Q[0], 10, 0.3, 0.0)
    put("l2_pref", idx_l2, pos_l2, Q[1], 7, 0.5, 0.2)
    # zero query: norm == 0 -> not normalised -> all sims exactly 0 (ids are a tie artefact: scores only)
    z = np.zeros(d, np.float32)
    ids, sc = run(idx, pos, z, 5, 0.3, 0.0)
    cases["zero_query"] = dict(q=z, k=5, eta=0.3, pref=0.0, ids=ids, scores=sc)
    # k > N raises inside NumPy (backends.py:468)
    try:
        idx.search(Q[0], k=n + 1, eta=0.3)
        exc = None
    except Exception as e:  # noqa: BLE001
        exc = (type(e).__name__, str(e))
    # k == 0 -> candidate_count <= 0 -> []
    empty = idx.search(Q[0], k=0, eta=0.3)
    meta = dict(n=n, d=d, k_gt_n_exception=exc, k_zero_len=len(empty))
    flat = {}
    for name, c in cases.items():
        for key, v in c.items():
            flat[f"{name}__{key}"] = np.asarray(v)
    np.savez_compressed(OUT / "g3_edge_cases.npz", E=E, stored_cos=np.asarray(idx._embeddings),
                        **{f"p_{key}": cols[key] for key in PKEYS}, **flat)
    (OUT / "g3_edge_cases.json").write_text(json.dumps(meta, indent=1))
    print("g3 ok", meta)


# ------------------------------------------------------------------ g4
def g4():
    out = {}
    meta = {}

    def fp32able(cols):
        return {k: np.asarray(v, np.float32).astype(np.float64) for k, v in cols.items()}

    def run_scorer(tag, cols, weights, delta, keys=orc.SIGNAL_KEYS):
        n = len(next(iter(cols.values())))
        rows = [{k: float(cols[k][i]) for k in keys} for i in range(n)]
        s = DewiScorer(weights=Weights(**weights) if weights else None, delta=delta)
        s.fit_stats(rows)
        out[f"{tag}__med"] = np.array([s.stats.medians[k] for k in keys])
        out[f"{tag}__mad"] = np.array([s.stats.mads[k] for k in keys])
        out[f"{tag}__score"] = np.array([s.score(r) for r in rows])
        out[f"{tag}__cond"] = np.array([s.score_conditional(r) for r in rows])
        for k in keys:
            out[f"{tag}__in_{k}"] = np.asarray(cols[k], np.float64)
        meta[tag] = dict(n=n, weights=weights, delta=delta, keys=list(keys))

    for n in (1, 2, 101, 10_000):
        cols = fp32able({k: v for k, v in orc.synth_payload_columns(n, seed=11 + n).items()})
        run_scorer(f"n{n}_default", cols, None, 3.0)
    cols = fp32able(orc.synth_payload_columns(101, seed=3))
    run_scorer("n101_weights", cols, dict(alpha_t=0.6, alpha_i=0.2, alpha_m=1.5, alpha_r=0.2, alpha_n=0.1), 3.0)
    run_scorer("n101_delta_small", cols, None, 0.5)  # most rows clip
    # 8 keys (rows built from Payload.to_dict(): 'dewi' is fitted too, tests/test_scorer_weights.py:9-11)
    run_scorer("n101_with_dewi_key", cols, None, 3.0, keys=PKEYS)
    # constant column -> MAD == 0 -> 1e-8
    cols0 = dict(cols)
    cols0["noise"] = np.full(101, 0.25)
    run_scorer("n101_mad_zero", cols0, None, 3.0)
    # the literal row of tests/test_scorer_weights.py:5-14 (f64 inputs that are NOT fp32-representable)
    w = Weights(alpha_t=0.6, alpha_i=0.2, alpha_r=0.2, alpha_n=0.1)
    s = DewiScorer(weights=w)
    sig = Payload(ht_mean=1.0, hi_mean=0.5, redundancy=0.2, noise=0.1, ht_q90=1.2, hi_q90=0.7).to_dict()
    sig["I_hat"] = 0.0
    s.fit_stats([sig])
    meta["literal_row"] = dict(sig=sig, score=s.score(sig), cond=s.score_conditional(sig),
                               medians=s.stats.medians, mads=s.stats.mads,
                               weights=dict(alpha_t=0.6, alpha_i=0.2, alpha_m=1.0, alpha_r=0.2, alpha_n=0.1))
    # ctor quirk: delta argument always overwrites weights.delta (scorer.py:37-40)
    meta["ctor_delta_quirk"] = DewiScorer(weights=Weights(delta=5.0)).weights.delta
    # even-N median = fp32 mean of the two middles
    rows = [{"x": v} for v in (0.1, 0.2, 0.4, 0.8)]
    from dewi.scorer import RobustStats
    st = RobustStats.fit(rows)
    meta["even_median"] = dict(values=[0.1, 0.2, 0.4, 0.8], med=st.medians["x"], mad=st.mads["x"])
    np.savez_compressed(OUT / "g4_scorer.npz", **out)
    (OUT / "g4_scorer.json").write_text(json.dumps(meta, indent=1))
    print("g4 ok")


# ------------------------------------------------------------------ g5
def g5():
    dim = 8
    d1 = OUT / "g5_exact_index"
    d2 = OUT / "g5_dewi_index"
    for d in (d1, d2):
        if d.exists():
            shutil.rmtree(d)
    ex = ExactIndex(dim=dim, space="cosine")
    di = DewiIndex(dim=dim, backend="auto", use_ann=False, rerank_eta=0.4, entropy_pref=0.1)
    cols = orc.synth_payload_columns(6, seed=21)
    for i in range(6):
        v = np.random.RandomState(42 + i).randn(dim).astype(np.float32)
        p = Payload(**{k: float(cols[k][i]) for k in PKEYS})
        ex.add(f"id-{i}", v, p)
        di.add(f"id-{i}", v, p, meta={"source": f"file{i}.txt"} if i % 2 == 0 else None)
    ex.build()
    di.build()
    ex.save(d1)
    di.save(d2)
    q = np.zeros(dim, np.float32)
    q[0] = 1.0
    r1 = ex.search(q, k=3, eta=0.5)
    r2 = di.search(q, k=3)
    (OUT / "g5_expected.json").write_text(json.dumps({
        "query": q.tolist(),
        "exact_k3_eta0.5": [[r[0], r[1]] for r in r1],
        "dewi_k3_defaults": [[r[0], r[1]] for r in r2],
    }, indent=1))
    print("g5 ok")


def g6():
    """Public API surface of the reference modules on the hot path: class -> public method -> parameters
    (name, kind, default).  Data only (names and default values), used by tests/test_cpu_host_logic.py to
    check that the drop-in package accepts every call the reference accepts."""
    import importlib
    import inspect
    surface = {}
    for modname in ("index", "backends", "scorer", "types"):
        mod = importlib.import_module("dewi." + modname)
        for cname, cls in vars(mod).items():
            if not (inspect.isclass(cls) and cls.__module__ == mod.__name__):
                continue
            members = {}
            for mname, member in vars(cls).items():
                if mname.startswith("_") and mname != "__init__":
                    continue
                fn = member.__func__ if isinstance(member, (classmethod, staticmethod)) else member
                if not callable(fn):
                    continue
                try:
                    sig = inspect.signature(fn)
                except (TypeError, ValueError):
                    continue
                members[mname] = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                                  for p in sig.parameters.values()]
            surface[f"{modname}.{cname}"] = members
    (OUT / "g6_api_surface.json").write_text(json.dumps(surface, indent=1, sort_keys=True))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    for name in which:
        globals()[name]()
    for p in sorted(OUT.rglob("*")):
        if p.is_file():
            print(f"{p.relative_to(REPO)}  {p.stat().st_size} B")
