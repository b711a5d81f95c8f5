"""Shared pytest plumbing: the ``gpu`` marker, import paths, golden loader.

Layout notes: the product package lives in the (hyphenated, hence not directly
importable) directory ``dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd/``;
we put that directory on ``sys.path`` so ``import dewi`` resolves to the
MI355X-native drop-in.  ``oracle/`` is put on the path for the tests only.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
PKG_DIR = REPO / "dewi-design-for-an-entropy-weighted-index-for-text-image-corpora_amd"
for p in (str(PKG_DIR), str(REPO / "oracle"), str(REPO)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: no silent fallbacks.
    if _has_gpu():
        return
    selected_gpu = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if selected_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
