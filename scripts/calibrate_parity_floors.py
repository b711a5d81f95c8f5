#!/usr/bin/env python3
"""CPU-only calibration of the decisive-query floors the GPU parity tests assert (tests/parity.py).

Whether a query is "decisive" depends on the inputs and the oracle alone (float64 gaps), not on the GPU
result, so the share of decisive queries of every seeded test case can be counted here, without a GPU:

    python3 scripts/calibrate_parity_floors.py

The floors in tests/test_hip_search.py, test_hip_bf16.py and test_hip_mfma.py sit a little below these
counts (stored rows come from the device's normalisation kernel and can differ from the oracle's in the
last fp32 bit, which may move a borderline query across the gap).
"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "tests"))
import numpy as np  # noqa: E402

import dewi_oracle as orc  # noqa: E402
import parity  # noqa: E402


def soa(cols):
    return orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])


print("== dimension sweep")
for dim in [8, 10, 16, 100, 128, 256, 260, 512, 768, 1024, 1536, 2048]:
    n = 3001 if dim <= 1024 else 1500
    raw = orc.synth_corpus(n, dim, seed=dim); cols = orc.synth_payload_columns(n, seed=dim)
    Q = orc.synth_queries(5, dim, seed=dim + 1); E = orc.build_matrix(raw); d,e = soa(cols)
    print(dim, [parity.count_decisive(E,Q,d,e,k,eta,pref,"cosine") for k,eta,pref in ((1, 0.3, 0.0), (10, 0.3, 0.0), (10, 0.7, -0.5), (100, 0.25, 0.3), (150, 0.5, 0.0))])
print("== l2")
for dim in [16,100,768]:
    rs = np.random.RandomState(dim); raw = (rs.randn(2000, dim) * 0.5).astype(np.float32)
    cols = orc.synth_payload_columns(2000, seed=3); Q = (rs.randn(5, dim) * 0.5).astype(np.float32); d,e=soa(cols)
    print(dim, parity.count_decisive(raw,Q,d,e,10,0.3,0.0,"l2"))
print("== eight queries")
for space in ["cosine","l2"]:
  for dim in [256,768,1536]:
    rs = np.random.RandomState(dim); scale = 0.5 if space == "l2" else 1.0
    raw = (rs.randn(3001, dim) * scale).astype(np.float32); cols = orc.synth_payload_columns(3001, seed=dim)
    Q = (rs.randn(21, dim) * scale).astype(np.float32); d,e=soa(cols)
    E = orc.build_matrix(raw) if space=="cosine" else raw
    print(space, dim, parity.count_decisive(E,Q,d,e,10,0.3,0.1,space), "/21")
def rnd_cases(n_cases, seed):
    rs = np.random.RandomState(seed)
    dims = [1, 3, 7, 8, 24, 33, 64, 96, 100, 128, 129, 200, 256, 384, 512, 640, 768, 1000]
    out = []
    for _ in range(n_cases):
        n = int(rs.choice([1, 2, 3, 17, 64, 255, 256, 257, 1000, 2049, 4000]))
        dim = int(rs.choice(dims)); b = int(rs.randint(1, 10)); k = int(rs.randint(1, min(n, 300) + 1))
        eta = float(rs.choice([0.0, 0.3, 0.5, 1.0, rs.rand()])); pref = float(rs.choice([0.0, -1.0, 0.5, rs.uniform(-1, 1)]))
        space = str(rs.choice(["cosine", "cosine", "l2"]))
        out.append((n, dim, b, k, eta, pref, space))
    return out
print("== fp32 random")
tot=dec=0
for case in rnd_cases(36, 20261004):
    n, dim, b, k, eta, pref, space = case
    rs = np.random.RandomState(n * 31 + dim)
    raw = (rs.randn(n, dim) * (0.5 if space == "l2" else 1.0)).astype(np.float32)
    cols = orc.synth_payload_columns(n, seed=dim)
    Q = (rs.randn(b, dim) * (0.5 if space == "l2" else 1.0)).astype(np.float32)
    with np.errstate(all='ignore'):
        E = orc.build_matrix(raw) if space=="cosine" else raw
    d,e=soa(cols)
    nd = parity.count_decisive(E,Q,d,e,k,eta,pref,space)
    tot+=b; dec+=nd
    print(case[:4], case[6], f"eta={eta:.2f} pref={pref:.2f}", nd, "/", b)
print(dec, tot)
def rnd_bf16(n_cases, seed):
    rs = np.random.RandomState(seed); out = []
    for _ in range(n_cases):
        n = int(rs.choice([1, 2, 3, 5, 64, 255, 257, 1001, 3000])); dim = int(rs.choice([8, 9, 40, 100, 128, 256, 300, 512, 768]))
        b = int(rs.randint(1, 10)); k = int(rs.randint(1, min(n, 120) + 1))
        out.append((n, dim, b, k, float(rs.choice([0.0, 0.3, 1.0])), float(rs.choice([0.0, 0.4, -1.0]))))
    return out
print("== bf16 random, gap 2e-5 / 4e-6")
tot=dec=dec2=0
for case in rnd_bf16(24, 4102026):
    n, dim, b, k, eta, pref = case
    raw = orc.synth_corpus(n, dim, seed=n * 7 + dim); cols = orc.synth_payload_columns(n, seed=n * 7 + dim)
    Eb = orc.bf16_round(orc.build_matrix(raw)); d,e=soa(cols)
    Q = orc.synth_queries(b, dim, seed=dim + b); Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q])
    nd = parity.count_decisive(Eb,Qp,d,e,k,eta,pref,"cosine",gap=2e-5,prepared=True)
    nd2 = parity.count_decisive(Eb,Qp,d,e,k,eta,pref,"cosine",gap=4e-6,prepared=True)
    tot+=b; dec+=nd; dec2+=nd2
    print(case, nd, nd2, "/", b)
print(dec, dec2, tot)
print("== bf16 fixed")
for dim,n in [(768, 4001), (768, 4000), (512, 3000), (256, 2501), (1024, 1501), (136, 2000), (100, 2000), (1280, 900)]:
    raw = orc.synth_corpus(n, dim, seed=dim+n); cols = orc.synth_payload_columns(n, seed=dim+n)
    Eb = orc.bf16_round(orc.build_matrix(raw)); d,e=soa(cols)
    Q = orc.synth_queries(5, dim, seed=dim); Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q])
    print(dim,n,[parity.count_decisive(Eb,Qp,d,e,k,eta,pref,"cosine",gap=2e-5,prepared=True) for k,eta,pref in ((10, 0.3, 0.0), (1, 0.5, 0.0), (100, 0.25, 0.3), (150, 0.5, 0.0))])
G = 1e-6
def rnd_bf16(n_cases, seed):
    rs = np.random.RandomState(seed); out = []
    for _ in range(n_cases):
        n = int(rs.choice([1, 2, 3, 5, 64, 255, 257, 1001, 3000])); dim = int(rs.choice([8, 9, 40, 100, 128, 256, 300, 512, 768]))
        b = int(rs.randint(1, 10)); k = int(rs.randint(1, min(n, 120) + 1))
        out.append((n, dim, b, k, float(rs.choice([0.0, 0.3, 1.0])), float(rs.choice([0.0, 0.4, -1.0]))))
    return out
for case in rnd_bf16(24, 4102026):
    n, dim, b, k, eta, pref = case
    raw = orc.synth_corpus(n, dim, seed=n * 7 + dim); cols = orc.synth_payload_columns(n, seed=n * 7 + dim)
    Eb = orc.bf16_round(orc.build_matrix(raw)); d,e=soa(cols)
    Q = orc.synth_queries(b, dim, seed=dim + b); Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q])
    nd = parity.count_decisive(Eb,Qp,d,e,k,eta,pref,"cosine",gap=G,prepared=True)
    if nd<b: print(case, nd, "/", b)
print("== bf16 fixed")
for dim,n in [(768, 4001), (768, 4000), (512, 3000), (256, 2501), (1024, 1501), (136, 2000), (100, 2000), (1280, 900)]:
    raw = orc.synth_corpus(n, dim, seed=dim+n); cols = orc.synth_payload_columns(n, seed=dim+n)
    Eb = orc.bf16_round(orc.build_matrix(raw)); d,e=soa(cols)
    Q = orc.synth_queries(5, dim, seed=dim); Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q])
    print(dim,n,[parity.count_decisive(Eb,Qp,d,e,k,eta,pref,"cosine",gap=G,prepared=True) for k,eta,pref in ((10, 0.3, 0.0), (1, 0.5, 0.0), (100, 0.25, 0.3), (150, 0.5, 0.0))])
print("== mfma cases")
for dim,n,b,k in [(768, 70_001, 256, 100), (768, 66_000, 40, 10), (512, 80_000, 300, 10), (256, 70_000, 17, 100), (128, 131_072, 64, 10)]:
    raw = orc.synth_corpus(n, dim, seed=dim+b); cols = orc.synth_payload_columns(n, seed=dim+b)
    Eb = orc.bf16_round(raw); d,e=soa(cols)
    Q = orc.synth_queries(b, dim, seed=b)[:32]; Qp = np.stack([orc.bf16_round(orc.prepare_query(q)) for q in Q])
    print(dim,n,b,k, parity.count_decisive(Eb,Qp,d,e,k,0.3,0.1,"cosine",gap=G,prepared=True,exact_gaps=False), "/", len(Q))
