"""Round-5 debugging aid: the C-ABI binning calls on a small scene, every persistent mode, status words after each call."""
import ctypes
import math
import os
import sys

os.environ.setdefault("GSR_BIN_NOTRAP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402

lib = dgr.lib
dev = torch.device("cuda:0")


def status():
    out = (ctypes.c_uint32 * 3)()
    lib.gsr_bin_persist_status(out)
    return f"status done={out[0]} code={out[1]:#x} faults={out[2]}"


def scene(N, W, H, sc, seed, ci):
    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc, device=dev)
    cam = S.orbit_cameras(4, W, H, device=dev)[ci]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                           1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                           False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mask = torch.ones(gy * gx, dtype=torch.uint8, device=dev)
    return (m2, depths, radii, co, mask), W, H


for cfg in [(2000, 200, 120, 0.01, 3, 1), (10000, 256, 256, 0.004, 0, 0), (20000, 979, 546, 0.006, 5, 3),
            (300000, 1920, 1080, None, 0, 0)]:
    N, W, H, sc, seed, ci = cfg
    if sc is None:
        g = S.make_gaussians(N, W, H, seed=seed, device=dev)
        cam = S.orbit_cameras(8, W, H, device=dev)[ci]
        rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                               torch.zeros(3, device=dev), 1.0, cam.world_view_transform,
                                               cam.full_proj_transform, 3, cam.camera_center, False, False)
        with torch.no_grad():
            m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
                g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
        gx, gy = (W + 15) // 16, (H + 15) // 16
        v = (m2, depths, radii, co, torch.ones(gy * gx, dtype=torch.uint8, device=dev))
    else:
        v, W, H = scene(*cfg)
    ref = None
    for mode in ("off", "prepare", "sort", "both"):
        dgr.set_bin_persistent(mode)
        dgr.release_workspaces()
        dgr.set_speculative_sort(False)
        for it in range(3):
            print(f"N {N} mode {mode} it {it} ...", end="", flush=True)
            pl, rg, D = dgr.bin_gaussians(*v, W, H)
            torch.cuda.synchronize()
            if ref is None:
                ref = (pl.clone(), rg.clone(), D)
            ok = D == ref[2] and torch.equal(rg, ref[1]) and torch.equal(pl[:D], ref[0][:D])
            print(f" D={D} {'ok' if ok else 'DIFFERS'}  {status()}", flush=True)
print("done")
