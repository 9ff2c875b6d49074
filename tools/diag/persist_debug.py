"""Round-5 aid: bin_gaussians (the operator's path: one C-ABI call per view, speculative bounded sort) in every
gsr_set_bin_persistent mode on the bench scene, lists compared with the look-back pipeline's.  GPU box only."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda:0")
g = S.make_gaussians(N, W, H, seed=0, device=dev)
cams = S.orbit_cameras(8, W, H, device=dev)
views = []
for cam in cams[:4]:
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                           1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                           False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    views.append((m2, depths, radii, co))
gx, gy = (W + 15) // 16, (H + 15) // 16
mask = torch.ones(gy * gx, dtype=torch.uint8, device=dev)
ref = {}
for mode in ("off", "prepare", "sort", "both"):
    dgr.set_bin_persistent(mode)
    for spec in (False, True):
        dgr.release_workspaces()
        dgr.set_speculative_sort(spec)
        for it in range(3):
            for k, v in enumerate(views):
                print(f"mode {mode} spec {spec} it {it} view {k} ...", end="", flush=True)
                pl, rg, D = dgr.bin_gaussians(*v, mask, W, H)
                torch.cuda.synchronize()
                if mode == "off" and not spec and it == 0:
                    ref[k] = (pl.clone(), rg.clone(), D)
                ok = D == ref[k][2] and torch.equal(rg, ref[k][1]) and torch.equal(pl[:D], ref[k][0][:D])
                print(f" D={D} {'ok' if ok else 'DIFFERS'}", flush=True)
print("done")
