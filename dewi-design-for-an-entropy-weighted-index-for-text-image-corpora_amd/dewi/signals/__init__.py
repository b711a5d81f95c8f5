"""Embedding-space signals for DEWI on the GPU.

The reference's ``dewi.signals`` estimators run HuggingFace models (GPT-2, ViT-MAE, CLIP) and are
outside the hot path this package rebuilds; the five estimator names are exported as ``None``
placeholders, exactly what the reference's own ``signals/__init__.py:11-34`` does when an optional
dependency is missing.  What IS provided is the arithmetic those estimators perform AFTER the
embedding models, on embeddings the caller already has:

* ``cross_modal_similarity`` — the ``I_hat`` signal: row-wise ``F.cosine_similarity`` of the text and
  image embeddings of the same document (reference ``signals/cross_modal.py:69, 124-139``).
* ``redundancy_top1`` — a per-document reduction of the text x image similarity matrix the reference's
  ``RedundancyEstimator`` returns (``signals/redundancy.py:28-39``).  The reference never defines
  such a reduction (``pipelines.py:148-149`` calls methods that do not exist), so this one —
  similarity of a document's text to the closest image of any OTHER document — is this package's
  own definition: parity unpinned.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _native as nat

TextEntropyEstimator = None
ImageEntropyEstimator = None
CrossModalDependency = None
RedundancyEstimator = None
NoiseEstimator = None


def cross_modal_similarity(text_emb, image_emb, return_device: bool = False, out=None):
    """``I_hat[i] = cos(text_emb[i], image_emb[i])`` for [N, d] fp32 arrays (host or CUDA tensors).
    ``return_device=True``: the fp32 [N] result stays on the GPU (a CUDA tensor, nothing synchronised) — ready to be a
    row of the signal table ``DewiScorer.fit_stats_columns`` / ``score_batch_device`` read in place; ``out`` (a CUDA
    fp32 [N] tensor, e.g. that row itself) receives the result instead of a fresh tensor."""
    import torch
    lib = nat.load_library()
    a = text_emb if isinstance(text_emb, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(text_emb, dtype=np.float32))
    b = image_emb if isinstance(image_emb, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image_emb, dtype=np.float32))
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError(f"expected two [N, d] arrays of the same shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    dev = a.device if a.is_cuda else (b.device if b.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    a = a.to(device=dev, dtype=torch.float32).contiguous()
    b = b.to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(a.shape[0], dtype=torch.float32, device=dev)
        elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (a.shape[0],)):
            raise ValueError("out must be a contiguous CUDA fp32 tensor of N elements")
        nat.check(lib.dewi_row_cosine_f32(nat.ptr(a), nat.ptr(b), nat.ptr(out), a.shape[0], a.shape[1], nat.stream_ptr()))
    return out if return_device else out.cpu().numpy()


def redundancy_top1(text_emb, image_emb, batch: int = 1024, bf16: Optional[bool] = None, return_device: bool = False):
    """Highest cosine similarity between document i's text and the image of any other document.

    Self-join through the kNN kernels: the normalised image embeddings are the corpus, the text
    embeddings the queries, k = 2 (the best match that is not the document itself).  ``bf16``
    (default: for 64K+ rows) stores the corpus in bf16 and uses the batched matrix-core path.
    Host arrays or CUDA tensors; ``return_device=True`` leaves the fp32 [N] result on the GPU.
    """
    import torch
    from .._engine import DeviceCorpus
    on_dev = nat.is_device_tensor(text_emb) and nat.is_device_tensor(image_emb)
    if on_dev:
        t_dev = text_emb.to(dtype=torch.float32).contiguous()
        if tuple(text_emb.shape) != tuple(image_emb.shape) or text_emb.dim() != 2:
            raise ValueError(f"expected two [N, d] arrays of the same shape, got {tuple(text_emb.shape)} and {tuple(image_emb.shape)}")
        n = int(t_dev.shape[0])
        t = None
    else:
        t = np.ascontiguousarray(text_emb.cpu().numpy() if hasattr(text_emb, "cpu") else text_emb, dtype=np.float32)
        im = np.ascontiguousarray(image_emb.cpu().numpy() if hasattr(image_emb, "cpu") else image_emb, dtype=np.float32)
        if t.shape != im.shape or t.ndim != 2:
            raise ValueError(f"expected two [N, d] arrays of the same shape, got {t.shape} and {im.shape}")
        n = t.shape[0]
    if n < 2:
        z = np.zeros(n, np.float32)
        return torch.from_numpy(z).cuda() if return_device else z
    if on_dev:
        lib = nat.load_library()
        emb = image_emb.to(dtype=torch.float32).contiguous().clone()      # the corpus: a normalised copy, on the device
        with torch.cuda.device(emb.device):
            nat.check(lib.dewi_normalize_rows_f32(nat.ptr(emb), nat.ptr(emb), n, int(emb.shape[1]), nat.stream_ptr()))
        zero = torch.zeros(n, dtype=torch.float32, device=emb.device)
        corpus = DeviceCorpus(emb, zero, zero, "cosine")
    else:
        zeros = np.zeros(n)
        corpus = DeviceCorpus.from_host(im, zeros, zeros, zeros, "cosine")
    if bf16 is None:
        bf16 = n >= 65536
    if bf16:
        corpus = corpus.to_bf16()
        if on_dev:
            del emb
    # Queries and results stay on the device: batches are enqueued back to back without a host round trip
    # (1 M x 512: the 977 batches are then bound by the matrix-core scans, ~1 PFLOP of them).
    dev = corpus.device
    if not on_dev:
        t_dev = torch.from_numpy(t).to(dev)
    out_dev = torch.empty(n, dtype=torch.float32, device=dev)
    rows = torch.arange(n, device=dev)
    for s in range(0, n, batch):
        e = min(n, s + batch)
        ids, sims = corpus.search_device(t_dev[s:e], 2, 0.0, 0.0)     # eta = 0: adjusted score == similarity
        own = ids[:, 0] == rows[s:e]
        out_dev[s:e] = torch.where(own, sims[:, 1], sims[:, 0])
        # (always answered: a query the matrix-core pass refuses is repaired inside the library call, ABI 5)
    return out_dev if return_device else out_dev.cpu().numpy()


__all__ = ["TextEntropyEstimator", "ImageEntropyEstimator", "CrossModalDependency", "RedundancyEstimator",
           "NoiseEstimator", "cross_modal_similarity", "redundancy_top1"]
