"""GPU tests of the embedding-space signals.

I_hat is pinned by the library call the reference makes: torch.nn.functional.cosine_similarity
(signals/cross_modal.py:69), evaluated here on the CPU.  redundancy_top1 is this package's own
definition (parity unpinned): checked against a NumPy brute force.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d", [(1000, 512), (4097, 512), (3, 512), (257, 10), (64, 768), (5, 3)])
def test_cross_modal_similarity_matches_torch(n, d):
    import torch
    import torch.nn.functional as F
    from dewi.signals import cross_modal_similarity
    rs = np.random.RandomState(n + d)
    t = rs.randn(n, d).astype(np.float32)
    im = (0.5 * t + rs.randn(n, d)).astype(np.float32)
    t[0] = 0.0                      # zero vector -> torch's eps clamp -> 0
    im[1] = t[1] * 1e-12            # tiny norm: exercises max(||x||, 1e-8)
    want = F.cosine_similarity(torch.from_numpy(t), torch.from_numpy(im), dim=1).numpy()
    got = cross_modal_similarity(t, im)
    assert got.shape == (n,) and got.dtype == np.float32
    assert np.allclose(got, want, rtol=0, atol=2e-6), np.abs(got - want).max()
    assert got[0] == 0.0


def test_redundancy_top1_vs_bruteforce():
    from dewi.signals import redundancy_top1
    rs = np.random.RandomState(3)
    n, d = 3000, 512
    t = rs.randn(n, d).astype(np.float32)
    im = (t + 0.8 * rs.randn(n, d)).astype(np.float32)
    im[10] = im[20]                                    # two documents share an image
    tn = t / np.linalg.norm(t, axis=1, keepdims=True)
    imn = im / np.linalg.norm(im, axis=1, keepdims=True)
    sim = tn.astype(np.float64) @ imn.astype(np.float64).T
    np.fill_diagonal(sim, -np.inf)
    want = sim.max(axis=1)
    got = redundancy_top1(t, im, batch=512)
    assert np.allclose(got, want, atol=2e-6)
    got16 = redundancy_top1(t, im, batch=512, bf16=True)       # bf16 corpus: coarser, same neighbours
    assert np.allclose(got16, want, atol=5e-3)


def test_reference_estimator_names_are_placeholders():
    import dewi.signals as sig
    assert sig.TextEntropyEstimator is None and sig.CrossModalDependency is None and sig.RedundancyEstimator is None
