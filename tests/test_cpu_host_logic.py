"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the host-side data contract behaves like the reference's, and the product
path refuses to run without a GPU instead of falling back."""
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from dewi import _native as nat
    header = (REPO / "include" / "dewi_hip.h").read_text()
    declared = set(re.findall(r"\b(dewi_[a-z0-9_]+)\s*\(", header))
    assert declared == set(nat.EXPORTED_SYMBOLS), declared ^ set(nat.EXPORTED_SYMBOLS)
    lib = nat.load_library(require_gpu=False)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dewi_abi_version() == nat.ABI_VERSION
    assert ctypes.sizeof(nat.DewiCandidate) == 16


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dewi import _native as nat
    from dewi.index import ExactIndex, Payload
    from dewi.scorer import DewiScorer
    idx = ExactIndex(dim=4)
    idx.add("a", np.ones(4, np.float32), Payload())
    with pytest.raises(nat.NativeLibraryError):
        idx.build()
    with pytest.raises(nat.NativeLibraryError):
        idx.search(np.ones(4, np.float32), k=1)
    with pytest.raises(nat.NativeLibraryError):
        DewiScorer().fit_stats([{"x": 1.0}])


def test_product_package_never_imports_the_oracle():
    pkg = REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"
    for path in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.hpp")):
        text = path.read_text()
        assert "dewi_oracle" not in text and "oracle/" not in text, path


def test_payload_serialization():
    """tests/test_index.py:73-101 of the reference."""
    from dewi.index import Payload
    p = Payload(dewi=0.5, ht_mean=1.2, ht_q90=1.8, hi_mean=0.8, hi_q90=1.2, I_hat=0.6, redundancy=0.1, noise=0.05)
    d = p.to_dict()
    assert isinstance(d, dict) and "dewi" in d and list(d) == ["dewi", "ht_mean", "ht_q90", "hi_mean", "hi_q90",
                                                               "I_hat", "redundancy", "noise"]
    assert Payload.from_dict(d).dewi == p.dewi
    b = p.to_bytes()
    assert isinstance(b, bytes) and Payload.from_bytes(b) == p
    assert Payload.from_dict({"dewi": "0.25", "extra": 1}) == Payload(dewi=0.25)     # extras ignored, float() cast
    assert Payload() == Payload(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


def test_payload_codec_matches_reference_files(golden_dir):
    from dewi.types import Payload
    lines = (golden_dir / "g5_exact_index" / "payloads.jsonl").read_text().splitlines()
    for line in lines:
        rec = json.loads(line)
        p = Payload.from_dict(rec["payload"])
        assert json.dumps({"doc_id": rec["doc_id"], "payload": p.to_dict()}) == line


def test_weights_defaults_and_columns():
    from dewi.types import Payload, Weights, payload_columns, payloads_from_columns
    w = Weights()
    assert (w.alpha_t, w.alpha_i, w.alpha_m, w.alpha_r, w.alpha_n, w.delta) == (1.0, 1.0, 1.0, 1.0, 1.0, 3.0)
    assert w.as_vector().tolist() == [1.0] * 5
    ps = [Payload(dewi=i / 10, ht_mean=i, hi_mean=2 * i) for i in range(5)]
    cols = payload_columns(ps)
    assert cols["hi_mean"].tolist() == [0, 2, 4, 6, 8] and cols["dewi"].dtype == np.float64
    assert payloads_from_columns(cols) == ps


def test_exact_index_host_side_contract(tmp_path):
    """Everything ExactIndex/DewiIndex do before touching the device."""
    from dewi.index import DewiIndex, ExactIndex, IndexBackend, Payload
    idx = ExactIndex(dim=4, space="cosine")
    assert idx.dim == 4 and idx.space == "cosine" and idx._doc_ids == [] and idx._payloads == {} and not idx._is_trained
    with pytest.raises(ValueError, match=r"Expected embedding of shape \(4,\), got \(5,\)"):
        idx.add("x", np.zeros(5, np.float32), Payload())
    p = Payload(dewi=0.3)
    idx.add("a", np.arange(4, dtype=np.float64), p)
    assert idx._doc_ids == ["a"] and idx._payloads["a"] is p and idx._embeddings[0].dtype == np.float32
    di = DewiIndex(dim=4, backend="definitely-not-a-backend")
    assert isinstance(di._backend, ExactIndex) and di.rerank_eta == 0.25 and di.entropy_pref == 0.0
    assert IndexBackend.from_str("exact") is IndexBackend.EXACT and len(di) == 0
    # empty index persists in the reference's format
    ExactIndex(dim=3).save(tmp_path / "e")
    meta = json.loads((tmp_path / "e" / "metadata.json").read_text())
    assert meta == {"dim": 3, "space": "cosine", "doc_ids": [], "normalize": True, "is_trained": False,
                    "num_embeddings": 0}
    assert not (tmp_path / "e" / "embeddings.npy").exists()


def test_loading_reference_saved_index_is_host_only(golden_dir):
    from dewi.index import DewiIndex, ExactIndex
    ex = ExactIndex.load(golden_dir / "g5_exact_index")
    assert ex.dim == 8 and ex._doc_ids == [f"id-{i}" for i in range(6)] and len(ex._payloads) == 6
    assert isinstance(ex._embeddings, np.ndarray) and ex._embeddings.shape == (6, 8)
    di = DewiIndex.load(golden_dir / "g5_dewi_index")
    assert len(di) == 6 and di.get_metadata("id-2") == {"source": "file2.txt"} and di.get_metadata("id-1") is None
    assert np.allclose(np.linalg.norm(di.get_embedding("id-4")), 1.0, atol=1e-6)


def test_public_api_accepts_every_reference_call(golden_dir):
    """g6: every public class / method of the reference's index, backends, scorer and types modules exists here
    with the same leading parameters (name, kind, default); extra trailing optional parameters are allowed."""
    import importlib
    import inspect
    surface = json.loads((golden_dir / "g6_api_surface.json").read_text())
    for qual, members in surface.items():
        modname, cname = qual.split(".")
        cls = getattr(importlib.import_module("dewi." + modname), cname, None)
        assert cls is not None, f"missing class {qual}"
        for mname, ref_params in members.items():
            assert hasattr(cls, mname), f"missing {qual}.{mname}"
            member = inspect.getattr_static(cls, mname)
            fn = member.__func__ if isinstance(member, (classmethod, staticmethod)) else member
            if not callable(fn):
                continue
            try:
                mine = list(inspect.signature(fn).parameters.values())
            except (TypeError, ValueError):
                continue
            kinds = {p.kind.name for p in mine}
            for i, (pname, kind, default) in enumerate(ref_params):
                if kind in ("VAR_POSITIONAL", "VAR_KEYWORD"):
                    assert kind in kinds, f"{qual}.{mname}: no {kind}"
                    continue
                assert i < len(mine) and mine[i].name == pname, f"{qual}.{mname}: parameter {i} should be {pname}"
                got = None if mine[i].default is inspect.Parameter.empty else repr(mine[i].default)
                assert got == default, f"{qual}.{mname}({pname}): default {got} != {default}"


def test_payload_store_serves_objects_and_columns():
    """dewi.types.PayloadStore: the reference's doc_id -> Payload dict for corpora ingested as columns."""
    from dewi.types import Payload, PayloadStore
    st = PayloadStore()
    a = Payload(dewi=1.0)
    st["a"] = a
    st.add_columns(1, ["b", "c", "d"], {"dewi": np.array([.1, .2, .3]), "ht_mean": np.array([1., 2., 3.])})
    assert len(st) == 4 and "c" in st and "z" not in st and dict.__len__(st) == 1     # nothing materialised yet
    p = st.at_row(2, "c")
    assert p.dewi == .2 and p.ht_mean == 2.0 and p.noise == 0.0
    assert st["c"] is p and st.get("c") is p and st.at_row(2, "c") is p              # one object per document
    assert st["a"] is a and st.get("zz") is None and len(st) == 4
    with pytest.raises(KeyError):
        st["zz"]
    blocks = list(st.column_blocks())
    assert blocks[0][0] == 1 and blocks[0][1] == 4 and set(blocks[0][3]) == {2}
    assert sorted(st.keys()) == ["a", "b", "c", "d"] and len(st) == 4 and len(list(st.items())) == 4
    with pytest.raises(ValueError, match="shape"):
        st.add_columns(4, ["e"], {"dewi": np.zeros(3)})


def test_column_ingest_host_side_bookkeeping():
    """ExactIndex.add_batch_columns mixed with add(): row-ordered float64 payload columns for the device."""
    from dewi.backends import ExactIndex
    from dewi.types import Payload
    idx = ExactIndex(dim=4)
    idx.add("x", np.ones(4, np.float32), Payload(dewi=0.9, ht_mean=2.0, hi_mean=4.0))
    idx.add_batch_columns(["p", "q"], np.ones((2, 4), np.float32), {"dewi": np.array([.1, .2]), "hi_mean": np.array([6., 8.])})
    idx.add("y", np.ones(4, np.float32), Payload(dewi=0.5))
    cols = idx._payload_columns()
    assert cols["dewi"].tolist() == [0.9, 0.1, 0.2, 0.5]
    assert cols["ht_mean"].tolist() == [2.0, 0.0, 0.0, 0.0] and cols["hi_mean"].tolist() == [4.0, 6.0, 8.0, 0.0]
    idx._payloads["q"].dewi = 7.0                       # an object handed out for a column row wins afterwards
    assert idx._payload_columns()["dewi"].tolist() == [0.9, 0.1, 7.0, 0.5]
    assert idx._doc_ids == ["x", "p", "q", "y"] and len(idx._payloads) == 4


def test_blocking_search_is_serialised_per_corpus():
    """DeviceCorpus.search stages through per-instance buffers, so it must hold the corpus lock from
    staging to copy-out.  Checked without a GPU on a stand-in instance whose device steps are stubs that
    record whether the lock was held."""
    import threading
    import torch
    from dewi import _engine as eng
    c = eng.DeviceCorpus.__new__(eng.DeviceCorpus)
    c._lock = threading.RLock()
    c.device = torch.device("cpu")
    c.id_offset = 0
    c._io = {}
    held = []

    def owned():
        # an RLock held by THIS thread can be re-acquired; probing from another thread tells if it is held at all
        res = []

        def probe():
            got = c._lock.acquire(blocking=False)
            if got:
                c._lock.release()
            res.append(got)
        t = threading.Thread(target=probe)
        t.start()
        t.join()
        return not res[0]
    c.stage_queries = lambda q: (held.append(owned()), torch.zeros((1, 4)))[1]
    c.search_device = lambda *a, **kw: held.append(owned())

    class _NoDev:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    real_device, real_empty, real_sync = torch.cuda.device, torch.empty, torch.cuda.current_stream
    torch.cuda.device = lambda d: _NoDev()
    torch.empty = lambda *a, **kw: real_empty(*a, **{k: v for k, v in kw.items() if k not in ("device", "pin_memory")})

    class _S:
        def synchronize(self):
            held.append(owned())
    torch.cuda.current_stream = lambda: _S()
    try:
        ids, sc = c.search(np.zeros(4, np.float32), k=2)
    finally:
        torch.cuda.device, torch.empty, torch.cuda.current_stream = real_device, real_empty, real_sync
    assert ids.shape == (1, 2) and held and all(held), held
    assert not owned()                                   # released afterwards


def test_bench_plain_gpus_n_reaches_the_launcher():
    """`python bench.py --gpus 4` with no launcher around it must start its own ranks (child torch.distributed.run, nothing
    exec'ed, no GPU touched first) instead of exiting: DEWI_BENCH_LAUNCH_DRYRUN prints the command it would run."""
    import json
    import os
    import subprocess
    import sys
    repo = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["DEWI_BENCH_LAUNCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "3"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launcher"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(str(repo / "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "3"]          # the same command line reaches every rank


def test_profiles_readme_block_matches_the_committed_files():
    """profiles/r03/README.md and profiles/r04/README.md end with a block generated from bench_*.json / kernel_stats_*.csv (scripts/profiles_table.py):
    it must be what those files say now."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    for tag in ("r03", "r04"):
        rc = subprocess.run([sys.executable, str(repo / "scripts" / "profiles_table.py"), tag, "--check"], cwd=repo).returncode
        assert rc == 0, f"profiles/{tag}/README.md: run `python scripts/profiles_table.py {tag}`"
