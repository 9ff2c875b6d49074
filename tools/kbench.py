"""Kernel-level micro-benchmark of the render op on the bench.py scene (HIP events per C-ABI call).
Usage (GPU box): python tools/kbench.py [--gaussians N] [--width W --height H] [--iters K]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--scale-coef", type=float, default=0.004)
    ap.add_argument("--calib", action="store_true", help="also run a 256 MiB device copy (PMC byte calibration)")
    ap.add_argument("--opacity-logit-mean", type=float, default=0.0)
    ap.add_argument("--opacity-logit-std", type=float, default=2.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H = a.width, a.height
    g = S.make_gaussians(a.gaussians, W, H, seed=0, scale_coef=a.scale_coef, device=dev,
                         opacity_logit_mean=a.opacity_logit_mean, opacity_logit_std=a.opacity_logit_std)
    cam = S.orbit_cameras(8, W, H, device=dev)[a.view]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                           torch.zeros(3, device=dev), 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 3, cam.camera_center, False, False)
    rast = dgr.GaussianRasterizer(rs)
    gg = {k: v.requires_grad_(True) for k, v in g.items()}
    wgt = torch.rand(3, H, W, device=dev)
    dgr.set_timing_mode("off")
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            dgr.kernel_timer.reset()
            dgr.kernel_timer.enabled = True
        m2, rgb, co, radii, depths = rast.preprocess_gaussians(gg["means3D"], gg["scales"], gg["rotations"],
                                                               gg["shs"], gg["opacities"], {})
        img, D, _, nc = rast.render_gaussians(m2, co, rgb, depths, radii, None, None, {})
        (img * wgt).sum().backward()
        for v in gg.values():
            v.grad = None
    if a.calib:
        x = torch.rand(64 * 1024 * 1024, device=dev)  # 256 MiB
        y = torch.empty_like(x)
        for _ in range(3):
            torch.mul(x, 2.0, out=y)  # vectorised elementwise kernel: 256 MiB read (16 B/lane) + 256 MiB written
    torch.cuda.synchronize()
    vis = int((radii > 0).sum())
    print(f"N={a.gaussians} visible={vis} D={D} D/tile={D / (((W + 15) // 16) * ((H + 15) // 16)):.0f} "
          f"mean n_contrib={nc.float().mean().item():.1f} max={nc.max().item()}")
    for k, (n, ms) in dgr.kernel_timer.summary_ms().items():
        print(f"  {k:22s} {ms:8.4f} ms  x{n}")


if __name__ == "__main__":
    main()
