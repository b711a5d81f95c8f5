"""Sharded robust fit ("fit_stats shards too"): the exact global median / MAD of
signal columns whose rows are split across ranks.

CPU (gloo, world_size 2 and 3): the orchestration of dewi/sharded.py::ShardedRobustFit — phase /
pass order, which workspace regions are summed over ranks, global row count, empty shards — with a
NumPy stand-in for the HIP steps (test-side code), against the oracle's fit of the whole table.
GPU (`-m gpu`): the real kernels, one process (world 1 == dewi_robust_fit_f32) and two ranks
sharing cuda:0, against the single-device fit and the oracle.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"
for p in (str(PKG), str(REPO / "oracle"), str(REPO / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

KEYS = ("ht_mean", "ht_q90", "hi_mean", "hi_q90", "I_hat", "redundancy", "noise")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _table(n, seed, with_nan=False, ties=False):
    import dewi_oracle as orc
    cols = orc.synth_payload_columns(n, seed=seed)
    t = np.stack([np.asarray(cols[k], dtype=np.float32) for k in KEYS])
    if ties:
        t[1] = np.round(t[1] * 4) / 4            # heavy duplicates around the median
        t[3] = 0.25                              # constant column: MAD == 0 -> 1e-8 on the host
    if with_nan and n > 3:
        t[5, n // 3] = np.nan                    # NumPy: any NaN -> median NaN
    return t


def _expected(t):
    import dewi_oracle as orc
    med, mad = orc.robust_fit({k: t[j] for j, k in enumerate(KEYS)})
    return med, mad


# ---- NumPy stand-in for HipFitSteps (same region / pick contract as include/dewi_hip.h) ----------
def _ord(x):
    x = np.asarray(x, np.float32)
    u = (x + np.float32(0.0)).view(np.uint32)
    key = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)
    return np.where(np.isnan(x), np.uint32(0xFFFFFFFF), key)


def _unord(k):
    k = np.uint32(k)
    u = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else ~k
    return np.array([u], np.uint32).view(np.float32)[0]


class NumpyFitSteps:
    BITS, SHIFT = (11, 11, 10), (21, 10, 0)

    def __init__(self, table):
        self.t = np.asarray(table, np.float32)
        self.ns = self.t.shape[0]
        self.begin()

    def begin(self):
        p = 2 * self.ns
        self.h = {(ph, ps): torch.zeros(p * 2048, dtype=torch.int32) for ph in (0, 1) for ps in (0, 1, 2)}
        self.nan = {ph: torch.zeros(self.ns, dtype=torch.int32) for ph in (0, 1)}
        self.prefix = np.zeros((2, p), np.uint32)
        self.rank = np.zeros((2, p), np.int64)
        self.med = np.zeros(self.ns, np.float32)

    def hist(self, phase, pass_):
        h = self.h[(phase, pass_)].numpy().reshape(2 * self.ns, 2048)
        for s in range(self.ns):
            x = self.t[s]
            if phase == 1:
                x = np.abs(x - self.med[s])
            if pass_ == 0:
                self.nan[phase][s] += int(np.isnan(x).sum())
            key = _ord(x)
            for j in (0, 1):
                p = 2 * s + j
                if pass_ == 0:
                    sel = key
                else:
                    sel = key[(key >> np.uint32(self.SHIFT[pass_] + self.BITS[pass_])) == self.prefix[phase, p]]
                d = (sel >> np.uint32(self.SHIFT[pass_])) & np.uint32((1 << self.BITS[pass_]) - 1)
                h[p] += np.bincount(d, minlength=2048).astype(np.int32)

    def regions(self, phase, pass_):
        return [self.h[(phase, pass_)]] + ([self.nan[phase]] if pass_ == 0 else [])

    def pick(self, n_total, phase, pass_):
        h = self.h[(phase, pass_)].numpy().reshape(2 * self.ns, 2048).astype(np.int64)
        for p in range(2 * self.ns):
            rank = ((n_total // 2) if (p & 1) else ((n_total - 1) // 2)) if pass_ == 0 else self.rank[phase, p]
            cum = np.cumsum(h[p])
            b = int(np.searchsorted(cum, rank, side="right"))
            self.rank[phase, p] = rank - (cum[b - 1] if b else 0)
            self.prefix[phase, p] = (int(self.prefix[phase, p]) << self.BITS[pass_] | b) if pass_ else b

    def finish(self, n_total, phase):
        out = np.zeros(self.ns, np.float32)
        for s in range(self.ns):
            a, b = _unord(self.prefix[phase, 2 * s]), _unord(self.prefix[phase, 2 * s + 1])
            r = a if (n_total & 1) else np.float32(np.float32(a + b) * np.float32(0.5))
            out[s] = np.float32(np.nan) if int(self.nan[phase][s]) else r
        if phase == 0:
            self.med = out
        return out


def _cpu_worker(rank, world, port, cases, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dewi.sharded import ShardedRobustFit, shard_bounds
        out = {}
        for name, (n, seed, with_nan, ties) in cases.items():
            t = _table(n, seed, with_nan, ties)
            lo, hi = shard_bounds(n, world)[rank]
            fit = ShardedRobustFit(NumpyFitSteps(t[:, lo:hi]), hi - lo)
            assert fit.n_total == n
            out[name] = fit.fit()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


CASES = {
    "odd": (1001, 5, False, False),
    "even_ties": (1000, 6, False, True),
    "nan": (257, 7, True, False),
    "tiny": (2, 8, False, False),            # world 3: one rank holds an empty shard
    "single": (1, 9, False, False),
}


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fit_on_cpu_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cpu_worker, args=(world, _free_port(), CASES, ret), nprocs=world, join=True)
    assert len(ret) == world
    for name, (n, seed, with_nan, ties) in CASES.items():
        med_e, mad_e = _expected(_table(n, seed, with_nan, ties))
        for r in range(world):
            med, mad = ret[r][name]
            for j, k in enumerate(KEYS):
                assert np.array_equal(np.float32(med[j]), np.float32(med_e[k]), equal_nan=True), (name, r, k)
                want = mad_e[k]
                got = float(np.float64(mad[j])) or 1e-8
                assert (np.isnan(want) and np.isnan(got)) or got == want, (name, r, k, got, want)


def test_numpy_steps_alone_equal_the_oracle():
    """world 1, no process group: the stand-in itself is a correct radix select."""
    from dewi.sharded import ShardedRobustFit
    t = _table(4096, 11, ties=True)
    med, mad = ShardedRobustFit(NumpyFitSteps(t), t.shape[1]).fit()
    med_e, mad_e = _expected(t)
    for j, k in enumerate(KEYS):
        assert float(med[j]) == med_e[k] and (float(np.float64(mad[j])) or 1e-8) == mad_e[k]


def test_empty_corpus_is_refused():
    from dewi.sharded import ShardedRobustFit
    with pytest.raises(IndexError):
        ShardedRobustFit(NumpyFitSteps(np.zeros((7, 0), np.float32)), 0)


# ---- GPU ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,with_nan,ties", [(1, False, False), (2, False, False), (100_001, False, True),
                                             (250_000, True, False)])
def test_hip_steps_one_rank_equal_monolithic_fit(n, with_nan, ties):
    from dewi.scorer import RobustStats
    from dewi.sharded import HipFitSteps, ShardedRobustFit
    t = _table(n, 21, with_nan, ties)
    med, mad = ShardedRobustFit(HipFitSteps(torch.from_numpy(t).cuda()), n).fit()
    whole = RobustStats.fit_columns({k: t[j] for j, k in enumerate(KEYS)})
    med_e, mad_e = _expected(t)
    for j, k in enumerate(KEYS):
        got_mad = float(np.float64(mad[j])) or 1e-8
        for want_med, want_mad in ((whole.medians[k], whole.mads[k]), (med_e[k], mad_e[k])):
            assert np.array_equal(np.float64(med[j]), np.float64(want_med), equal_nan=True), (k, med[j], want_med)
            assert (np.isnan(want_mad) and np.isnan(got_mad)) or got_mad == want_mad, (k, got_mad, want_mad)


def _gpu_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dewi.scorer import DewiScorer
        from dewi.sharded import shard_bounds
        torch.cuda.set_device(0)
        out = {}
        for name, (n, seed, with_nan, ties) in {"big": (300_001, 33, False, True), "nan": (5000, 34, True, False),
                                                 "tiny": (1, 35, False, False)}.items():
            t = _table(n, seed, with_nan, ties)
            lo, hi = shard_bounds(n, world)[rank]
            sc = DewiScorer()
            sc.fit_stats_sharded({k: t[j, lo:hi] for j, k in enumerate(KEYS)})
            out[name] = (dict(sc.stats.medians), dict(sc.stats.mads))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_fit_equals_oracle():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for name, (n, seed, with_nan, ties) in {"big": (300_001, 33, False, True), "nan": (5000, 34, True, False),
                                             "tiny": (1, 35, False, False)}.items():
        med_e, mad_e = _expected(_table(n, seed, with_nan, ties))
        for r in (0, 1):
            med, mad = ret[r][name]
            for k in KEYS:
                assert np.array_equal(np.float64(med[k]), np.float64(med_e[k]), equal_nan=True), (name, r, k)
                assert (np.isnan(mad_e[k]) and np.isnan(mad[k])) or mad[k] == mad_e[k], (name, r, k)


@pytest.mark.gpu
def test_c5_eight_shards_on_one_gpu_full_size():
    """Config C5's fit at its full size (1M documents x 7 signals) as EIGHT doc-id shards, replayed in one process
    on one GPU: every shard runs the real histogram steps on its 125 K rows (its own workspace), the regions the
    ranks would all-reduce over RCCL are summed here with torch, every shard picks with the global row count.  All
    eight must arrive at the oracle's medians / MADs bit for bit — and at the single-device two-launch fit's.
    (The wire itself — RCCL with 8 ranks — is not exercised: multi-GPU is unmeasured on hardware.)"""
    from dewi.scorer import RobustStats
    from dewi.sharded import HipFitSteps, shard_bounds
    n, world = 1_000_000, 8
    t = _table(n, 55, with_nan=False, ties=True)
    steps = [HipFitSteps(torch.from_numpy(np.ascontiguousarray(t[:, lo:hi])).cuda()) for lo, hi in shard_bounds(n, world)]
    for st in steps:
        st.begin()
    out = []
    for phase in (0, 1):
        for pass_ in (0, 1, 2):
            for st in steps:
                st.hist(phase, pass_)
            regions = [st.regions(phase, pass_) for st in steps]
            for which in range(len(regions[0])):                    # what dist.all_reduce(SUM) does on 8 ranks
                total = torch.stack([r[which] for r in regions]).sum(dim=0, dtype=torch.int32)
                for r in regions:
                    r[which].copy_(total)
            for st in steps:
                st.pick(n, phase, pass_)
        out.append([st.finish(n, phase).cpu().numpy().copy() for st in steps])
    med_e, mad_e = _expected(t)
    whole = RobustStats.fit_columns({k: t[j] for j, k in enumerate(KEYS)})
    for r in range(world):
        for j, k in enumerate(KEYS):
            med, mad = float(np.float64(out[0][r][j])), float(np.float64(out[1][r][j])) or 1e-8
            assert med == med_e[k] == whole.medians[k], (r, k, med, med_e[k])
            assert mad == mad_e[k] == whole.mads[k], (r, k, mad, mad_e[k])
