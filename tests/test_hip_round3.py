"""GPU tests added in round 3, all through the C ABI.

* the doc-id sharded path answers every k the single device answers: shard candidate records beyond 2048 per
  shard (dewi_knn_candidates / dewi_knn_scan + dewi_knn_finish sort in the workspace) and merges of more than
  2048 records per query (dewi_merge_rerank rank-merges the sorted shard lists through a workspace) — the
  reference has no such limit (backends.py:439-471);
"""
import numpy as np
import pytest

import dewi_oracle as orc
from parity import compare_query

pytestmark = pytest.mark.gpu


def _engine():
    from dewi import _engine
    return _engine


def _shards(eng, raw, cols, bounds, space="cosine", bf16=False):
    out = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        sub = {key: v[lo:hi] for key, v in cols.items()}
        sh = eng.DeviceCorpus.from_host(raw[lo:hi], sub["dewi"], sub["ht_mean"], sub["hi_mean"], space, id_offset=lo)
        out.append(sh.to_bf16() if bf16 else sh)
    return out


@pytest.mark.parametrize("n,bounds,k", [
    (9000, [0, 1000, 2300, 2310, 4000, 5500, 7000, 8000, 9000], 200),      # 8 shards x 400 records: 3200 > 2048 (LDS limit)
    (9000, [0, 1000, 2300, 2310, 4000, 5500, 7000, 8000, 9000], 1500),     # c = 3000: every shard is short of c
    (30000, [0, 9000, 19000, 30000], 1500),                                # c = 3000 < shard rows: large select per shard
    (30000, [0, 9000, 19000, 30000], 1025),                                # smallest k on the global-memory path
    (12000, [0, 5000, 12000], 6000),                                       # c == n_total: everything is a candidate
    (4000, [0, 2500, 4000], 4000),                                         # k == n_total
])
def test_sharded_large_k_equals_single_device_and_oracle(n, bounds, k):
    """candidates per shard + merge == ONE search over the whole corpus, bit for bit, at candidate counts beyond what the
    select / merge kernels sort in LDS; and the answer is the oracle's (near-tie rules: with hundreds of results
    some adjacent pair is always closer than the gap)."""
    import torch
    eng = _engine()
    d, eta, pref = 64, 0.3, 0.1
    raw = orc.synth_corpus(n, d, seed=n + k)
    cols = orc.synth_payload_columns(n, seed=n + k)
    Q = orc.synth_queries(3, d, seed=k)
    whole = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    ids_w, sc_w = whole.search(Q, k, eta, pref)
    c = min(2 * k, n)
    qd = torch.from_numpy(Q).cuda()
    lists = torch.stack([sh.candidates_device(qd, c) for sh in _shards(eng, raw, cols, bounds)])
    recs = eng.records_to_numpy(lists)
    for s, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        real = min(c, hi - lo)
        assert np.all(recs["id"][s][:, :real] >= lo) and np.all(recs["id"][s][:, :real] < hi)
        assert np.all(recs["id"][s][:, real:] == -1)                              # padding behind a short shard's rows
        assert np.all(np.diff(recs["sim"][s][:, :real], axis=1) <= 0)              # sorted by similarity
    ids, sc = eng.merge_rerank_device(lists, c, k, eta, pref)
    assert np.array_equal(ids.cpu().numpy(), ids_w) and np.array_equal(sc.cpu().numpy(), sc_w)
    E = whole.emb.cpu().numpy()
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for j in range(Q.shape[0]):
        _, msg = compare_query(E, Q[j], dewi32, ent32, k, eta, pref, "cosine", ids_w[j], sc_w[j])
        assert msg is None, (j, msg)
        ref_ids, _ = orc.search(E, Q[j], dewi32, ent32, k, eta, pref)
        assert np.mean(ids_w[j] == ref_ids) > 0.98                                 # near-tie swaps only
        if k == n:
            assert sorted(ids_w[j].tolist()) == list(range(n))


def test_scan_finish_split_with_large_candidate_counts():
    """dewi_knn_scan + dewi_knn_finish (PipelinedSearcher, two streams) beyond 2048 candidates: final results and shard
    records equal the one-call entry points."""
    import torch
    eng = _engine()
    n, d, k = 20_000, 128, 1300
    raw = orc.synth_corpus(n, d, seed=5)
    cols = orc.synth_payload_columns(n, seed=5)
    c = eng.DeviceCorpus.from_host(raw, cols["dewi"], cols["ht_mean"], cols["hi_mean"], id_offset=70_000)
    Q = torch.from_numpy(orc.synth_queries(4, d, seed=6)).cuda()
    want_ids, want_sc = c.search_device(Q, k, 0.4, 0.0)
    pipe = eng.PipelinedSearcher(c, k, 0.4, 0.0, n_queries=2)
    ids = torch.empty((2, 2, k), dtype=torch.int64, device="cuda")
    sc = torch.empty((2, 2, k), dtype=torch.float32, device="cuda")
    for j in range(2):
        pipe.submit(Q[2 * j:2 * j + 2], ids[j], sc[j])
    pipe.drain()
    # dewi_knn_finish adds the shard's id offset, the one-call search returns local rows
    assert torch.equal(ids.view(4, k), want_ids + 70_000) and torch.equal(sc.view(4, k), want_sc)
    recs = torch.empty((2, 2, 2 * k, 4), dtype=torch.int32, device="cuda")
    pipe2 = eng.PipelinedSearcher(c, k, 0.4, 0.0, n_queries=2, n_candidates=2 * k)
    for j in range(2):
        pipe2.submit(Q[2 * j:2 * j + 2], out_records=recs[j])
    pipe2.drain()
    assert torch.equal(recs.view(4, 2 * k, 4), c.candidates_device(Q, 2 * k))
    assert int(recs[..., 3].min()) >= 70_000


def test_large_merge_carries_the_refusal_marker_and_rejects_a_short_workspace():
    """A shard whose batched path refused a query marks its records with id -2: the large merge hands the query back
    as unanswered (ids -1), like the LDS merge; and the C ABI checks the workspace it now needs."""
    import torch
    from dewi import _native as nat
    eng = _engine()
    n, d, k = 6000, 64, 600
    raw = orc.synth_corpus(n, d, seed=1)
    cols = orc.synth_payload_columns(n, seed=1)
    c = 2 * k
    qd = torch.from_numpy(orc.synth_queries(2, d, seed=2)).cuda()
    lists = torch.stack([sh.candidates_device(qd, c) for sh in _shards(eng, raw, cols, [0, 2000, 4000, 6000])])
    good_ids, _ = eng.merge_rerank_device(lists, c, k, 0.3, 0.0)
    marked = lists.clone()
    marked[1, 0, :, 3] = -2                                   # shard 1 refused query 0
    ids, sc = eng.merge_rerank_device(marked, c, k, 0.3, 0.0)
    assert torch.all(ids[0] == -1) and torch.all(torch.isnan(sc[0]))
    assert torch.equal(ids[1], good_ids[1])
    lib = nat.load_library()
    need = int(lib.dewi_merge_workspace_bytes(3, 2, c, c))
    assert need > 0 and int(lib.dewi_merge_workspace_bytes(3, 2, 20, 20)) == 0
    small = torch.empty(need - 8, dtype=torch.uint8, device="cuda")
    rc = lib.dewi_merge_rerank(nat.ptr(lists), 3, 2, c, c, k, 0.3, 0.0, nat.ptr(ids), nat.ptr(sc), nat.ptr(small), need - 8,
                               nat.stream_ptr())
    assert rc == nat.ERR_WORKSPACE and "workspace" in nat.last_error()
