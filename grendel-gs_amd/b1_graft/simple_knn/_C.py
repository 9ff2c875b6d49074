"""`simple_knn._C.distCUDA2` stand-in (scene/gaussian_model.py:20,163-166 of the reference): mean squared
distance of every point to its 3 nearest neighbours, used ONCE at initialisation to size the first
Gaussians.  Off the hot path ("init only"), so it is plain torch on the device: exact
brute-force k-NN in row chunks (O(N^2) distance evaluations; fine for SfM point clouds of 1e5-1e6 points)."""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    pts = points.detach().float().contiguous()
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    k = min(4, n)  # self + 3 neighbours
    sq = (pts * pts).sum(1)
    chunk = max(1, min(n, (1 << 28) // max(n, 1)))  # ~1 GiB of distances per chunk
    for s in range(0, n, chunk):
        q = pts[s:s + chunk]
        d2 = (sq[s:s + chunk, None] + sq[None, :] - 2.0 * (q @ pts.t())).clamp_(min=0.0)
        near = torch.topk(d2, k, dim=1, largest=False).values  # ascending; [:, 0] is the point itself
        out[s:s + chunk] = near[:, 1:].sum(1) / 3.0
    return out
