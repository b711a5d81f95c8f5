#!/usr/bin/env python3
"""Build-and-search harness with the command line and the metrics.json layout of the reference's
scripts/profile_index.py (reference scripts/profile_index.py:239-292), run against this package, so that
numbers are comparable with anything measured through the reference script:

    python3 scripts/profile_index.py --n-docs 100000 --dim 256 --n-queries 1000 --k 10 --output profile_results

metrics.json: {"build": {n_docs, dim, data_generation_time, index_construction_time, docs_per_second},
               "search": {n_queries, k, total_search_time, queries_per_second, latency_ms}}
`--per-row-add` builds with one DewiIndex.add call per document as the reference script does (Python-bound),
`--objects` with the bulk add_batch of Payload objects; the default is add_batch_columns (payloads as arrays:
no Python object per document; Payload objects are made for the rows a search returns).  Searches go one query at a time through DewiIndex.search (host query in,
[(doc_id, score, Payload)] out), like the reference's loop.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"))


def synthetic_corpus(n_docs, dim, seed=42, objects=True):
    """Distributions of the reference harness: unit-norm Gaussian rows, Beta / Gamma signals."""
    from dewi.types import Payload
    rng = np.random.RandomState(seed)
    emb = rng.randn(n_docs, dim).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    ids = [f"doc_{i:08d}" for i in range(n_docs)]
    cols = {
        "dewi": np.clip(rng.beta(2, 2, n_docs), 0, 1), "ht_mean": rng.gamma(2, 0.5, n_docs),
        "ht_q90": 1.5 * rng.gamma(2, 0.5, n_docs), "hi_mean": rng.gamma(2, 0.3, n_docs),
        "hi_q90": 1.5 * rng.gamma(2, 0.3, n_docs), "I_hat": rng.beta(2, 2, n_docs),
        "redundancy": rng.beta(1, 5, n_docs), "noise": rng.beta(1, 10, n_docs),
    }
    payloads = [Payload(**{k: float(v[i]) for k, v in cols.items()}) for i in range(n_docs)] if objects else cols
    return ids, emb, payloads


def build(ids, emb, payloads, mode):
    from dewi.index import DewiIndex
    index = DewiIndex(dim=emb.shape[1], use_ann=False, rerank_eta=0.3)
    if mode == "per_row":
        for i, doc_id in enumerate(ids):
            index.add(doc_id, emb[i], payloads[i])
    elif mode == "objects":
        index.add_batch(ids, emb, payloads)
    else:
        index.add_batch_columns(ids, emb, payloads)
    index.build()
    return index


def main():
    ap = argparse.ArgumentParser(description="Profile DEWI index build and search on the MI355X path")
    ap.add_argument("--n-docs", type=int, default=100000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--n-queries", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--output", type=str, default="profile_results")
    ap.add_argument("--skip-build", action="store_true")
    ap.add_argument("--skip-search", action="store_true")
    ap.add_argument("--per-row-add", action="store_true", help="one add() per document, as the reference script")
    ap.add_argument("--objects", action="store_true", help="bulk add_batch of Payload objects instead of columns")
    a = ap.parse_args()
    out_dir = Path(a.output)
    out_dir.mkdir(parents=True, exist_ok=True)
    metrics = {}

    t0 = time.time()
    mode = "per_row" if a.per_row_add else ("objects" if a.objects else "columns")
    ids, emb, payloads = synthetic_corpus(a.n_docs, a.dim, objects=mode != "columns")
    gen_time = time.time() - t0
    t0 = time.time()
    index = build(ids, emb, payloads, mode)
    build_time = time.time() - t0
    if not a.skip_build:
        metrics["build"] = {"n_docs": a.n_docs, "dim": a.dim, "data_generation_time": gen_time,
                            "index_construction_time": build_time, "docs_per_second": a.n_docs / build_time,
                            "ingest": mode}
    if not a.skip_search:
        rng = np.random.RandomState(7)
        queries = rng.randn(a.n_queries, a.dim).astype(np.float32)
        queries /= np.linalg.norm(queries, axis=1, keepdims=True)
        for q in queries[:10]:
            index.search(q, k=a.k)
        t0 = time.time()
        for q in queries:
            index.search(q, k=a.k)
        search_time = time.time() - t0
        metrics["search"] = {"n_queries": a.n_queries, "k": a.k, "total_search_time": search_time,
                             "queries_per_second": a.n_queries / search_time,
                             "latency_ms": search_time / a.n_queries * 1000}
    (out_dir / "metrics.json").write_text(json.dumps(metrics, indent=2))
    print(json.dumps(metrics, indent=2))
    print(f"Profile results saved to: {out_dir.absolute()}")


if __name__ == "__main__":
    main()
