"""Mint the frozen rasterizer golden vectors tests/golden/raster_*.npz (SURVEY.md 8(c) "golden vectors we must mint").

    python tests/golden/make_raster_golden.py            # rewrites every raster_*.npz

PARITY UNPINNED: the reference's rasterizer source is absent (empty submodule) and it ships no tests, so these
vectors are NOT outputs of the reference.  They freeze the outputs of THIS repo's float64 autograd oracle
(oracle/torch_oracle.py) on the 8(c) list of scenes, so that the two CPU restatements and the HIP kernels are all
held against one committed set of numbers and cannot drift together unnoticed.  Regenerate only on purpose.

Every file is self-contained: inputs (float32), camera, background, tile mask, SH degree and the float64 results
rounded to float32 (image, final_T, the five preprocess outputs, the gradient of
sum(image * golden_weight) with respect to the five inputs and to means2D); a float64 checksum of the inputs is stored
and checked by the loader.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import synthetic_scene as S  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

KEYS = ["means3D", "scales", "rotations", "shs", "opacities"]


def golden_weight(H, W):
    """analytic dL/dimage (float64): nothing random to store or to regenerate"""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    return torch.stack([0.5 + 0.5 * torch.sin(0.37 * x + 0.91 * y + 1.3 * c) for c in range(3)])


def checksum(g):
    return np.array([float(g[k].double().sum()) for k in KEYS] + [float(g[k].double().abs().sum()) for k in KEYS])


def _ident(n):
    return torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1)


def scenes():
    """name -> dict(g, cam, bg, mask(None = all), sh_degree, regen(None | kwargs of make_gaussians))"""
    out = {}

    cam = S.SyntheticCamera(0, 64, 48)
    shs = torch.zeros(1, 16, 3); shs[0, 0] = torch.tensor([1.2, 0.3, -0.4]); shs[0, 1:] = 0.05
    out["single"] = dict(g=dict(means3D=torch.tensor([[0.1, -0.05, 4.0]]), scales=torch.tensor([[0.2, 0.1, 0.15]]),
                                rotations=torch.nn.functional.normalize(torch.tensor([[0.9, 0.1, -0.3, 0.2]])), shs=shs,
                                opacities=torch.tensor([[0.7]])),
                         cam=cam, bg=torch.tensor([0.1, 0.2, 0.3]), sh_degree=3)

    shs = torch.zeros(2, 16, 3); shs[:, 0] = torch.tensor([[1.0, 0.2, 0.1], [-0.2, 0.9, 1.4]])
    shs[:, 1:] = torch.linspace(-0.1, 0.1, 90).view(2, 15, 3)
    out["two_overlap"] = dict(g=dict(means3D=torch.tensor([[0.05, 0.0, 3.0], [-0.1, 0.08, 5.0]]),
                                     scales=torch.tensor([[0.25, 0.12, 0.2], [0.5, 0.6, 0.3]]),
                                     rotations=torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.1, 0.0],
                                                                                           [0.3, -0.5, 0.4, 0.7]])),
                                     shs=shs, opacities=torch.tensor([[0.6], [0.85]])),
                              cam=cam, bg=torch.tensor([0.0, 0.0, 0.0]), sh_degree=3)

    # (inputs are stored: torch's CPU generator is NOT bit-reproducible across hosts -- the AVX2 / AVX-512 code paths of
    # randn differ -- so "regenerate from the seed" fails on the GPU box)
    out["rand10k_256"] = dict(g=S.make_gaussians(10000, 256, 256, seed=0, scale_coef=0.004),
                              cam=S.orbit_cameras(4, 256, 256)[0], bg=torch.zeros(3), sh_degree=3)

    W, H = 160, 96
    g = S.make_gaussians(300, W, H, seed=12, scale_coef=0.02)
    g["means3D"][::3, 2] = torch.linspace(-5.0, 0.25, 100)  # behind the camera / inside the 0.2 near plane
    out["behind_camera"] = dict(g=g, cam=S.SyntheticCamera(0, W, H), bg=torch.tensor([0.2, 0.2, 0.2]), sh_degree=3)

    g = S.make_gaussians(300, W, H, seed=13, scale_coef=0.02)
    g["opacities"][::2, 0] = torch.linspace(0.0005, 0.0045, 150)  # straddles 1/255 = 0.003922
    out["low_opacity"] = dict(g=g, cam=S.SyntheticCamera(0, W, H), bg=torch.tensor([0.0, 0.3, 0.6]), sh_degree=3)

    n = 300
    out["saturating"] = dict(
        g=dict(means3D=torch.cat([torch.zeros(n, 2), torch.linspace(2, 9, n)[:, None]], 1),
               scales=torch.full((n, 3), 0.3), rotations=_ident(n),
               shs=torch.rand(n, 16, 3, generator=torch.Generator().manual_seed(0)) * 0.5,
               opacities=torch.full((n, 1), 0.6)),
        cam=S.SyntheticCamera(0, 32, 32), bg=torch.zeros(3), sh_degree=3)

    W, H = 160, 128  # 8 tile rows, cut 4 | 4: big splats straddle the border and appear on both ranks
    g = S.make_gaussians(500, W, H, seed=14, scale_coef=0.03)
    for j, rows in enumerate([(0, 4), (4, 8)]):
        m = torch.zeros(8, 10, dtype=torch.bool); m[rows[0]:rows[1]] = True
        out[f"border_band{j}"] = dict(g=g, cam=S.orbit_cameras(4, W, H)[1], bg=torch.tensor([0.3, 0.1, 0.2]), mask=m,
                                      sh_degree=3)

    W, H = 160, 96
    g = S.make_gaussians(1500, W, H, seed=15, scale_coef=0.012, sh_rest_sigma=0.5)  # strong view dependence
    for d in range(4):
        out[f"sh{d}"] = dict(g=g, cam=S.orbit_cameras(4, W, H)[1], bg=torch.tensor([0.05, 0.1, 0.15]), sh_degree=d)
    out["white_bg"] = dict(g=g, cam=S.orbit_cameras(4, W, H)[2], bg=torch.ones(3), sh_degree=2)
    out["black_bg"] = dict(g=g, cam=S.orbit_cameras(4, W, H)[2], bg=torch.zeros(3), sh_degree=2)

    W, H = 48, 1080  # 67.5 tile rows: the ragged last row of a 1080-row image
    out["rows1080"] = dict(g=S.make_gaussians(4000, W, H, seed=16, scale_coef=0.01, fx=0.9 * H),
                           cam=S.SyntheticCamera(0, W, H, fx=0.9 * H), bg=torch.tensor([0.4, 0.4, 0.1]), sh_degree=3)
    return out


def run_oracle(sc, dtype=torch.float64):
    cam, g = sc["cam"], sc["g"]
    W, H = cam.image_width, cam.image_height
    gx, gy = O.tile_grid(W, H)
    mask = sc.get("mask")
    mask = torch.ones(gy, gx, dtype=torch.bool) if mask is None else mask
    kw = dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center, W=W,
              H=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=sc["sh_degree"])
    ins = {k: v.to(dtype).clone().requires_grad_(True) for k, v in g.items()}
    m2, rgb, co, radii, depths = O.preprocess(*[ins[k] for k in KEYS], **kw)
    m2.retain_grad()
    img, fT, nc = O.render(m2, co, rgb, depths, radii, mask, bg=sc["bg"].to(dtype), W=W, H=H)
    (img * golden_weight(H, W).to(dtype)).sum().backward()
    res = dict(image=img, final_T=fT, n_contrib=nc, means2D=m2, rgb=rgb, conic_opacity=co, radii=radii, depths=depths,
               d_means2D=m2.grad if m2.grad is not None else torch.zeros_like(m2))
    for k in KEYS:
        res["d_" + k] = ins[k].grad if ins[k].grad is not None else torch.zeros_like(ins[k])
    return {k: v.detach() for k, v in res.items()}, mask, kw


def main():
    for name, sc in scenes().items():
        res, mask, kw = run_oracle(sc)
        cam = sc["cam"]
        blob = dict(W=np.int32(kw["W"]), H=np.int32(kw["H"]), tanfovx=np.float64(kw["tanfovx"]),
                    tanfovy=np.float64(kw["tanfovy"]), sh_degree=np.int32(sc["sh_degree"]),
                    viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                    campos=cam.camera_center.numpy(), bg=sc["bg"].numpy(), mask=mask.numpy(),
                    input_checksum=checksum(sc["g"]))
        if sc.get("regen") is None:
            for k in KEYS:
                blob["in_" + k] = sc["g"][k].numpy()
        else:
            blob["regen"] = np.array([sc["regen"]["n"], sc["regen"]["width"], sc["regen"]["height"],
                                      sc["regen"]["seed"]], dtype=np.int64)
            blob["regen_scale_coef"] = np.float64(sc["regen"]["scale_coef"])
        for k, v in res.items():
            blob["out_" + k] = v.numpy() if v.dtype in (torch.int32, torch.int64) else v.to(torch.float32).numpy()
        path = os.path.join(HERE, f"raster_{name}.npz")
        np.savez_compressed(path, **blob)
        vis = int((res["radii"] > 0).sum())
        print(f"{name:14s} N={sc['g']['means3D'].shape[0]:6d} visible={vis:6d} {kw['W']}x{kw['H']} "
              f"img max {float(res['image'].max()):.3f} -> {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
