"""N4: `simple_knn._C.distCUDA2` as a HIP operator (gsr_knn_mean_dist2) against the brute-force C restatement --
bit-exact (both evaluate (dx*dx + dy*dy) + dz*dz without contraction and average the same three minima)."""
import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from oracle import cref as C

pytestmark = pytest.mark.gpu


def _cloud(n, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(n - n // 3, 3, generator=g) * torch.tensor([10.0, 4.0, 1.0])
    b = 0.02 * torch.randn(n // 3, 3, generator=g) + torch.tensor([5.0, 2.0, 0.5])  # dense cluster
    return torch.cat([a, b])[torch.randperm(n, generator=g)].contiguous()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 257, 1024, 1025, 5000, 40000])
def test_knn_matches_bruteforce_bit_exact(device, n):
    pts = _cloud(n, seed=n)
    out = dgr.knn_mean_dist2(pts.to(device)).cpu()
    ref = C.knn_mean_dist2(pts)
    assert torch.equal(out, ref), (out - ref).abs().max()


def test_knn_duplicates_planes_and_nonfinite(device):
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(3000, 3, generator=g)
    pts[:500] = pts[500:1000]          # exact duplicates: distance 0 neighbours count
    pts[1000:2000, 2] = 0.25           # a plane (degenerate extent in z for many points)
    out = dgr.knn_mean_dist2(pts.to(device)).cpu()
    assert torch.equal(out, C.knn_mean_dist2(pts))
    same = torch.full((300, 3), 1.5)   # zero-extent cloud
    assert torch.equal(dgr.knn_mean_dist2(same.to(device)).cpu(), torch.zeros(300))
    bad = pts.clone()
    bad[7, 1] = float("nan")
    bad[9, 0] = float("inf")
    out = dgr.knn_mean_dist2(bad.to(device)).cpu()
    ref = C.knn_mean_dist2(bad)
    keep = torch.ones(3000, dtype=torch.bool)
    keep[[7, 9]] = False
    assert torch.equal(out[keep], ref[keep])


def test_knn_1m_points_against_kdtree_sample(device):
    """full size (an SfM cloud of 1e6 points): exactness on a random sample against an independent k-d tree"""
    from scipy.spatial import cKDTree

    n = 1_000_000
    pts = _cloud(n, seed=11)
    out = dgr.knn_mean_dist2(pts.to(device)).cpu().double().numpy()
    tree = cKDTree(pts.double().numpy())
    idx = np.random.default_rng(0).choice(n, 20000, replace=False)
    d, _ = tree.query(pts[idx].double().numpy(), k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    assert np.allclose(out[idx], ref, rtol=2e-4, atol=1e-10)
    assert np.isfinite(out).all() and (out >= 0).all()


def test_graft_module_resolves_to_operator(device):
    import os
    import sys

    graft = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "grendel-gs_amd", "b1_graft")
    sys.path.insert(0, graft)
    try:
        from simple_knn._C import distCUDA2
    finally:
        sys.path.remove(graft)
    pts = _cloud(2000, seed=5)
    assert torch.equal(distCUDA2(pts.to(device)).cpu(), C.knn_mean_dist2(pts))
