"""Pin the CPU oracle (oracle/dewi_oracle.py) against outputs of the REAL reference.

tests/golden/*.npz were produced by oracle/gen_golden.py, which imports the
reference from /root/reference/src in the build container.  Everything here is
bit-exact (ids, fp32 scores, fp32 medians/MADs) except the float64 scorer
output, which is allowed 2 ulp because np.exp's vector and scalar code paths may
round differently.
"""
import json

import numpy as np
import pytest

import dewi_oracle as orc

PKEYS = orc.PAYLOAD_KEYS


def _cols(g):
    return {k: g[f"p_{k}"] for k in PKEYS}


def test_g1_test_index_shape(golden):
    g = golden("g1_test_index_shape.npz")
    E = orc.build_matrix(g["E"])
    assert np.array_equal(E, g["stored"])                      # A1/A2 bit-exact
    cols = _cols(g)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    k = int(g["k"])
    for a, eta in enumerate(g["etas"]):
        for b, pref in enumerate(g["prefs"]):
            for j, q in enumerate(g["Q"]):
                ids, sc = orc.search(E, q, dewi32, ent32, k, float(eta), float(pref))
                assert np.array_equal(ids, g["ids"][a, b, j]), (eta, pref, j)
                assert np.array_equal(sc, g["scores"][a, b, j]), (eta, pref, j)


def test_g2_c1_10k_768(golden):
    g = golden("g2_c1_10k_768.npz")
    n, d, k, eta = int(g["n"]), int(g["d"]), int(g["k"]), float(g["eta"])
    raw = orc.synth_corpus(n, d, seed=int(g["corpus_seed"]))
    Q = orc.synth_queries(g["ids"].shape[0], d, seed=int(g["query_seed"]))
    cols = orc.synth_payload_columns(n, seed=int(g["corpus_seed"]))
    # the regenerated inputs are the ones the reference saw
    assert raw.astype(np.float64).sum() == float(g["e_sum"])
    assert Q.astype(np.float64).sum() == float(g["q_sum"])
    assert cols["dewi"].sum() == float(g["dewi_sum"])
    E = orc.build_matrix(raw)
    assert np.array_equal(E[:4], g["stored_rows_0_3"])
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    for j in range(Q.shape[0]):
        ids, sc, cand, csim = orc.search(E, Q[j], dewi32, ent32, k, eta, 0.0, return_candidates=True)
        assert np.array_equal(ids, g["ids"][j])
        assert np.array_equal(sc, g["scores"][j])
        o = np.argsort(-csim, kind="stable")
        assert np.array_equal(cand[o], g["cand_ids"][j])
        assert np.array_equal(csim[o], g["cand_sims"][j])


def test_g3_edge_cases(golden, golden_dir):
    g = golden("g3_edge_cases.npz")
    meta = json.loads((golden_dir / "g3_edge_cases.json").read_text())
    cols = _cols(g)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    E_cos = orc.build_matrix(g["E"], "cosine")
    assert np.array_equal(E_cos, g["stored_cos"])
    E_l2 = orc.build_matrix(g["E"], "l2")
    assert np.array_equal(E_l2, g["E"])                          # l2 stores raw fp32 rows
    names = sorted({key.split("__")[0] for key in g.files if "__" in key})
    assert len(names) == 10
    for name in names:
        q, k = g[f"{name}__q"], int(g[f"{name}__k"])
        eta, pref = float(g[f"{name}__eta"]), float(g[f"{name}__pref"])
        space = "l2" if name.startswith("l2") else "cosine"
        E = E_l2 if space == "l2" else E_cos
        ids, sc = orc.search(E, q, dewi32, ent32, k, eta, pref, space)
        assert np.array_equal(sc, g[f"{name}__scores"]), name
        assert np.array_equal(ids, g[f"{name}__ids"]), name
    # k > N: the reference raises ValueError out of NumPy (backends.py:468)
    assert meta["k_gt_n_exception"][0] == "ValueError"
    with pytest.raises(ValueError):
        orc.search(E_cos, g["k_eq_n__q"], dewi32, ent32, meta["n"] + 1, 0.3)
    assert meta["k_zero_len"] == 0
    assert orc.search(E_cos, g["k_eq_n__q"], dewi32, ent32, 0, 0.3)[0].size == 0
    with pytest.raises(ValueError):
        orc.build_matrix(np.empty((0, 4), np.float32))


def test_g4_scorer(golden, golden_dir):
    g = golden("g4_scorer.npz")
    meta = json.loads((golden_dir / "g4_scorer.json").read_text())
    tags = [t for t in meta if isinstance(meta[t], dict) and "keys" in meta[t]]
    assert len(tags) == 8
    for tag in tags:
        m = meta[tag]
        keys = m["keys"]
        cols = {k: g[f"{tag}__in_{k}"] for k in keys}
        med, mad = orc.robust_fit(cols)
        assert np.array_equal(np.array([med[k] for k in keys]), g[f"{tag}__med"]), tag
        assert np.array_equal(np.array([mad[k] for k in keys]), g[f"{tag}__mad"]), tag
        w = m["weights"]
        wv = None if w is None else [w["alpha_t"], w["alpha_i"], w["alpha_m"], w["alpha_r"], w["alpha_n"]]
        for mode, key in (("standard", "score"), ("conditional", "cond")):
            got = orc.score(cols, med, mad, wv, m["delta"], mode)
            ref = g[f"{tag}__{key}"]
            assert np.allclose(got, ref, rtol=5e-16, atol=0), (tag, mode, np.abs(got - ref).max())
    # literal tests/test_scorer_weights.py row: f64 inputs, fp32 fit -> 0.3027... not 0.5
    lit = meta["literal_row"]
    cols = {k: np.array([v]) for k, v in lit["sig"].items()}
    med, mad = orc.robust_fit(cols)
    assert med == lit["medians"] and mad == lit["mads"]
    w = lit["weights"]
    wv = [w["alpha_t"], w["alpha_i"], w["alpha_m"], w["alpha_r"], w["alpha_n"]]
    assert orc.score(cols, med, mad, wv, 3.0, "standard")[0] == pytest.approx(lit["score"], rel=1e-15)
    assert orc.score(cols, med, mad, wv, 3.0, "conditional")[0] == pytest.approx(lit["cond"], rel=1e-15)
    assert lit["score"] == pytest.approx(0.30275610348537835, rel=1e-15)
    ev = meta["even_median"]
    med, mad = orc.robust_fit({"x": np.array(ev["values"])})
    assert med["x"] == ev["med"] == 0.30000001192092896
    assert mad["x"] == ev["mad"] == 0.15000000596046448
    assert meta["ctor_delta_quirk"] == 3.0


def test_decision_gaps_are_sane(golden):
    g = golden("g1_test_index_shape.npz")
    cols = _cols(g)
    dewi32, ent32 = orc.payload_soa(cols["dewi"], cols["ht_mean"], cols["hi_mean"])
    cut, rank = orc.decision_gaps(g["stored"], g["Q"][0], dewi32, ent32, 10, 0.3)
    assert cut > 0 and rank > 0
