"""The reference's own index tests (tests/test_index.py, test_index_delegation.py,
test_index_roundtrip.py), restated against the MI355X drop-in: same fixtures, same
assertions, plus persistence interchange with directories the reference itself saved."""
import json

import numpy as np

import dewi_oracle as orc
import pytest

pytestmark = pytest.mark.gpu

DIM, N_DOCS, N_QUERIES, K = 128, 100, 5, 10


def _fixtures():
    from dewi.index import Payload
    rs = np.random.RandomState(42)
    embs = rs.randn(N_DOCS, DIM).astype(np.float32)
    embs /= np.linalg.norm(embs, axis=1, keepdims=True)
    pays = [Payload(dewi=float(np.clip(rs.beta(2, 2), 0, 1)), ht_mean=float(rs.gamma(2, 0.5)),
                    ht_q90=float(rs.gamma(2, 0.5) * 1.5), hi_mean=float(rs.gamma(2, 0.3)),
                    hi_q90=float(rs.gamma(2, 0.3) * 1.5), I_hat=float(rs.beta(2, 2)),
                    redundancy=float(rs.beta(1, 5)), noise=float(rs.beta(1, 10))) for _ in range(N_DOCS)]
    ids = [f"doc_{i}" for i in range(N_DOCS)]
    qs = rs.randn(N_QUERIES, DIM).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    return ids, embs, pays, qs


def test_exact_index():
    from dewi.index import ExactIndex, Payload
    ids, embs, pays, qs = _fixtures()
    index = ExactIndex(dim=DIM, space="cosine")
    for d, e, p in zip(ids, embs, pays):
        index.add(d, e, p)
    index.build()
    for q in qs:
        results = index.search(q, k=K)
        assert len(results) == K
        scores = [r[1] for r in results]
        assert all(scores[i] >= scores[i + 1] for i in range(len(scores) - 1))
        for doc_id, score, payload in results:
            assert doc_id in ids and isinstance(score, float) and isinstance(payload, Payload)
    with pytest.raises(ValueError, match="Expected embedding of shape"):
        index.add("bad", np.zeros(DIM + 1, np.float32), Payload())
    with pytest.raises(ValueError, match="No embeddings"):
        ExactIndex(dim=4).build()


def test_ann_backends_absent_like_reference_without_libs():
    from dewi.index import FAISSIndex, HNSWIndex, IndexBackend, _HAS_FAISS, _HAS_HNSW
    assert not _HAS_FAISS and not _HAS_HNSW
    with pytest.raises(ImportError):
        HNSWIndex(dim=8)
    with pytest.raises(ImportError):
        FAISSIndex(dim=8)
    assert IndexBackend.from_str("auto") is IndexBackend.EXACT
    with pytest.raises(KeyError):
        IndexBackend.from_str("nope")


def test_dewi_index_factory():
    from dewi.index import DewiIndex
    ids, embs, pays, qs = _fixtures()
    index = DewiIndex(dim=DIM, space="cosine")        # backend="auto", use_ann=True -> exact fallback
    for d, e, p in zip(ids[:20], embs[:20], pays[:20]):
        index.add(d, e, p)
    index.build()
    results = index.search(qs[0], k=K)
    assert 0 < len(results) <= K
    assert len(index) == 20 and index.get_payload("doc_3") is pays[3]
    assert np.allclose(index.get_embedding("doc_3"), embs[3], atol=1e-6) and index.get_embedding("nope") is None
    with pytest.raises(ValueError, match=r"Expected query shape \(128,\)"):
        index.search(np.zeros((1, DIM), np.float32))


def test_index_persistence(tmp_path):
    from dewi.index import ExactIndex, Payload
    ids, embs, pays, qs = _fixtures()
    index = ExactIndex(dim=DIM, space="cosine")
    for d, e, p in zip(ids[:10], embs[:10], pays[:10]):
        index.add(d, e, p)
    index.build()
    before = index.search(qs[0], k=5)
    save_dir = tmp_path / "test_index"
    index.save(save_dir)
    assert (save_dir / "metadata.json").exists() and (save_dir / "payloads.jsonl").exists()
    assert (save_dir / "embeddings.npy").exists()
    loaded = ExactIndex.load(save_dir)
    assert loaded.dim == DIM and len(loaded._doc_ids) == 10 and set(loaded._doc_ids) == set(ids[:10])
    for d in ids[:10]:
        assert d in loaded._payloads and isinstance(loaded._payloads[d], Payload)
    results = loaded.search(qs[0], k=5)
    assert len(results) == 5
    assert [r[0] for r in results] == [r[0] for r in before]
    assert [r[1] for r in results] == [r[1] for r in before]        # stored rows are not re-normalised


def test_entropy_preference():
    from dewi.index import ExactIndex, Payload
    index = ExactIndex(dim=DIM, space="cosine")
    n = 50
    rs = np.random.RandomState(42)
    embs = rs.randn(n, DIM).astype(np.float32)
    embs /= np.linalg.norm(embs, axis=1, keepdims=True)
    for i in range(n):
        ent = float(i / n)
        index.add(f"doc_{i}", embs[i], Payload(dewi=0.5, ht_mean=ent, ht_q90=ent * 1.5, hi_mean=ent, hi_q90=ent * 1.5,
                                               I_hat=0.5, redundancy=0.1, noise=0.05))
    index.build()
    q = np.ones(DIM, np.float32) / np.sqrt(DIM)

    def avg_ent(res):
        return np.mean([(r[2].ht_mean + r[2].hi_mean) / 2 for r in res])

    pos, neu, neg = (avg_ent(index.search(q, k=10, entropy_pref=p)) for p in (1.0, 0.0, -1.0))
    assert pos >= neu >= neg


def test_dewi_reranking():
    from dewi.index import ExactIndex, Payload
    index = ExactIndex(dim=DIM, space="cosine")
    n = 50
    rs = np.random.RandomState(42)
    embs = rs.randn(n, DIM).astype(np.float32)
    embs /= np.linalg.norm(embs, axis=1, keepdims=True)
    for i in range(n):
        index.add(f"doc_{i}", embs[i], Payload(dewi=float(i / n), ht_mean=1.0, ht_q90=1.5, hi_mean=0.8, hi_q90=1.2,
                                               I_hat=0.5, redundancy=0.1, noise=0.05))
    index.build()
    q = np.ones(DIM, np.float32) / np.sqrt(DIM)

    def avg_dewi(res):
        return np.mean([r[2].dewi for r in res])

    hi, mid, lo = (avg_dewi(index.search(q, k=10, eta=e)) for e in (1.0, 0.5, 0.0))
    assert hi >= mid >= lo


def test_index_delegation_shapes_and_no_error():
    from dewi.index import DewiIndex
    from dewi.types import Payload
    dim = 16
    idx = DewiIndex(dim=dim, backend="auto", use_ann=False)
    for j in range(10):
        idx.add(f"id-{j}", np.random.RandomState(100 + j).randn(dim).astype(np.float32), Payload(dewi=0.5))
    idx.build()
    q = np.zeros(dim, np.float32)
    q[0] = 1.0
    assert len(idx.search(q, k=5, eta=0.0, entropy_pref=0.0)) == 5


def test_index_roundtrip(tmp_path):
    from dewi.index import DewiIndex, Payload
    dim = 8
    idx = DewiIndex(dim=dim, backend="auto", use_ann=False)
    for i in range(5):
        idx.add(f"id-{i}", np.random.RandomState(42 + i).randn(dim).astype(np.float32), payload=Payload())
    idx.build()
    q = np.zeros(dim, np.float32)
    q[0] = 1.0
    res = idx.search(q, k=3)
    assert len(res) == 3
    idx.save(tmp_path / "idx")
    re = DewiIndex.load(tmp_path / "idx")
    res2 = re.search(q, k=3)
    assert [r[0] for r in res2] == [r[0] for r in res]


def test_loads_directories_saved_by_the_reference(golden_dir):
    """G5: ExactIndex.save / DewiIndex.save output of the reference itself loads and searches."""
    from dewi.index import DewiIndex, ExactIndex
    exp = json.loads((golden_dir / "g5_expected.json").read_text())
    q = np.array(exp["query"], np.float32)
    ex = ExactIndex.load(golden_dir / "g5_exact_index")
    got = ex.search(q, k=3, eta=0.5)
    assert [r[0] for r in got] == [r[0] for r in exp["exact_k3_eta0.5"]]
    assert np.allclose([r[1] for r in got], [r[1] for r in exp["exact_k3_eta0.5"]], atol=1e-6)
    di = DewiIndex.load(golden_dir / "g5_dewi_index")
    assert di.rerank_eta == 0.4 and di.entropy_pref == 0.1 and di.get_metadata("id-0") == {"source": "file0.txt"}
    got = di.search(q, k=3)
    assert [r[0] for r in got] == [r[0] for r in exp["dewi_k3_defaults"]]
    assert np.allclose([r[1] for r in got], [r[1] for r in exp["dewi_k3_defaults"]], atol=1e-6)


def _g5_inputs():
    """The seeded inputs oracle/gen_golden.py g5() fed the reference (ids, raw rows, payloads, metadata)."""
    from dewi.types import Payload
    dim = 8
    cols = orc.synth_payload_columns(6, seed=21)
    keys = ("dewi", "ht_mean", "ht_q90", "hi_mean", "hi_q90", "I_hat", "redundancy", "noise")
    rows = [np.random.RandomState(42 + i).randn(dim).astype(np.float32) for i in range(6)]
    pays = [Payload(**{k: float(cols[k][i]) for k in keys}) for i in range(6)]
    metas = [{"source": f"file{i}.txt"} if i % 2 == 0 else None for i in range(6)]
    return dim, [f"id-{i}" for i in range(6)], rows, pays, metas


def _assert_same_saved_index(mine, ref):
    """metadata.json equal as JSON incl. key order, payloads.jsonl line for line, embeddings.npy same dtype / shape /
    C order and at most 2 ulp per element: the reference divides by ``np.linalg.norm`` — an fp32 BLAS dot product, up to
    ~1.5 ulp off the true norm in an order NumPy does not specify — where the device divides by the float64-summed norm
    rounded once (<= 0.5 ulp); each quotient adds its own half ulp.  (Measured on these 48 values: 39 equal, 8 one ulp,
    1 two ulp.)"""
    a, b = json.loads((mine / "metadata.json").read_text()), json.loads((ref / "metadata.json").read_text())
    assert a == b and list(a) == list(b), (a, b)
    assert (mine / "payloads.jsonl").read_text().splitlines() == (ref / "payloads.jsonl").read_text().splitlines()
    ea, eb = np.load(mine / "embeddings.npy", allow_pickle=False), np.load(ref / "embeddings.npy", allow_pickle=False)
    assert ea.dtype == eb.dtype == np.float32 and ea.shape == eb.shape and ea.flags["C_CONTIGUOUS"] and not np.isfortran(ea)
    ulp = np.spacing(np.abs(eb).astype(np.float32))
    assert np.all(np.abs(ea - eb) <= 2 * ulp), float(np.max(np.abs(ea - eb) / ulp))
    assert sorted(p.name for p in mine.iterdir()) == sorted(p.name for p in ref.iterdir())


@pytest.mark.parametrize("bulk", [False, True], ids=["add", "add_batch"])
def test_saves_what_the_reference_saves(golden_dir, tmp_path, bulk):
    """G5, write side (reference backends.py:483-513, index.py:121-146): the same seeded inputs ingested here (one-row
    ``add`` as the reference's generator did, or the bulk path), built on the GPU and saved, against the directories the
    reference itself wrote — file names, JSON keys and their order, payload lines, the embedding matrix."""
    from dewi.index import DewiIndex, ExactIndex
    dim, ids, rows, pays, metas = _g5_inputs()
    ex = ExactIndex(dim=dim, space="cosine")
    di = DewiIndex(dim=dim, backend="auto", use_ann=False, rerank_eta=0.4, entropy_pref=0.1)
    if bulk:
        ex.add_batch(ids, np.stack(rows), pays)
        di.add_batch(ids[:3], np.stack(rows[:3]), pays[:3])          # bulk blocks and single rows mixed
        for i in range(3, 6):
            di.add(ids[i], rows[i], pays[i], meta=metas[i])
        di._meta.update({ids[i]: metas[i] for i in range(3) if metas[i] is not None})   # (add_batch takes no metadata)
        di._meta = {k: di._meta[k] for k in sorted(di._meta)}
    else:
        for i in range(6):
            ex.add(ids[i], rows[i], pays[i])
            di.add(ids[i], rows[i], pays[i], meta=metas[i])
    ex.build()
    di.build()
    ex.save(tmp_path / "ex")
    di.save(tmp_path / "di")
    _assert_same_saved_index(tmp_path / "ex", golden_dir / "g5_exact_index")
    ref = golden_dir / "g5_dewi_index"
    mine = tmp_path / "di"
    for name in ("config.json", "meta.json"):
        a, b = json.loads((mine / name).read_text()), json.loads((ref / name).read_text())
        assert a == b and list(a) == list(b), (name, a, b)
    assert sorted(p.name for p in mine.iterdir()) == sorted(p.name for p in ref.iterdir())
    _assert_same_saved_index(mine / "ann_index", ref / "ann_index")
    # and the reference's own answers on its saved index come back from ours
    exp = json.loads((golden_dir / "g5_expected.json").read_text())
    q = np.array(exp["query"], np.float32)
    got = ExactIndex.load(tmp_path / "ex").search(q, k=3, eta=0.5)
    assert [r[0] for r in got] == [r[0] for r in exp["exact_k3_eta0.5"]]
    assert np.allclose([r[1] for r in got], [r[1] for r in exp["exact_k3_eta0.5"]], atol=1e-6)


def test_batch_api_equals_single_queries():
    from dewi.index import DewiIndex
    ids, embs, pays, qs = _fixtures()
    index = DewiIndex(dim=DIM, use_ann=False)
    index.add_batch(ids, embs, pays)
    batched = index.search_batch(qs, k=K, eta=0.3)
    for q, res in zip(qs, batched):
        single = index.search(q, k=K, eta=0.3)
        assert [(r[0], r[1]) for r in single] == [(r[0], r[1]) for r in res]
        assert all(a[2] is b[2] for a, b in zip(single, res))
    index.add("late", embs[0], pays[0])            # add after build -> lazily rebuilt
    assert index.search(embs[0], k=2, eta=0.0)[0][0] in ("doc_0", "late")
    assert len(index) == N_DOCS + 1


@pytest.mark.parametrize("space", ["cosine", "l2"])
def test_batch_api_on_the_matrix_core_pass_equals_single_queries(space):
    """DewiIndex.search_batch at a size where a batch takes the depth-split matrix-core pass (>= 64 K rows, >= 5
    queries; cosine and l2) against DewiIndex.search one query at a time (row-per-wave kernels): same documents up to
    near-tie swaps, scores equal to fp32 rounding, the same Payload objects."""
    from dewi.index import DewiIndex
    n, dim, b, k = 70_000, 256, 12, 10
    rs = np.random.RandomState(3)
    embs = rs.randn(n, dim).astype(np.float32)
    cols = {"dewi": np.clip(rs.beta(2, 2, n), 0, 1), "ht_mean": rs.gamma(2, 0.5, n), "ht_q90": rs.gamma(2, 0.5, n) * 1.5,
            "hi_mean": rs.gamma(2, 0.3, n), "hi_q90": rs.gamma(2, 0.3, n) * 1.5, "I_hat": rs.beta(2, 2, n),
            "redundancy": rs.beta(1, 5, n), "noise": rs.beta(1, 10, n)}
    index = DewiIndex(dim=dim, space=space, use_ann=False)
    index.add_batch_columns([f"doc_{i}" for i in range(n)], embs, cols)
    qs = rs.randn(b, dim).astype(np.float32)
    batched = index.search_batch(qs, k=k, eta=0.3)
    agree = 0
    for q, res in zip(qs, batched):
        single = index.search(q, k=k, eta=0.3)
        agree += sum(a[0] == r[0] for a, r in zip(single, res))
        scale = max(1.0, max(abs(r[1]) for r in res))
        assert np.allclose(sorted(a[1] for a in single), sorted(r[1] for r in res), rtol=0, atol=3e-6 * scale)
        for a, r in zip(single, res):
            if a[0] == r[0]:
                assert a[2] is r[2]
    assert agree >= 0.97 * b * k


@pytest.mark.parametrize("flag", [None, "--objects", "--per-row-add"])
def test_profile_harness_writes_the_reference_metrics_layout(tmp_path, flag):
    """scripts/profile_index.py — the reference harness's command line (reference scripts/profile_index.py:239-292)
    on this package — run as the reference's users run it: one child process, metrics.json with the reference's
    sections and keys, every ingest mode (columns, Payload objects, one add() per document)."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(repo / "scripts" / "profile_index.py"), "--n-docs", "3000", "--dim", "128",
           "--n-queries", "40", "--k", "10", "--output", str(tmp_path / "prof")]
    if flag:
        cmd.append(flag)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = json.loads((tmp_path / "prof" / "metrics.json").read_text())
    assert set(m) == {"build", "search"}
    assert {"n_docs", "dim", "data_generation_time", "index_construction_time", "docs_per_second"} <= set(m["build"])
    assert set(m["search"]) == {"n_queries", "k", "total_search_time", "queries_per_second", "latency_ms"}
    assert m["build"]["n_docs"] == 3000 and m["search"]["n_queries"] == 40 and m["search"]["queries_per_second"] > 0
