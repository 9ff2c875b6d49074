import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch
import diff_gaussian_rasterization as dgr
import synthetic_scene as S
dev = torch.device("cuda:0")
N, W, H, sc, seed, ci = 2000, 200, 120, 0.01, 3, 1
g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc, device=dev)
cam = S.orbit_cameras(4, W, H, device=dev)[ci]
rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                       1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
with torch.no_grad():
    m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
gx, gy = (W + 15) // 16, (H + 15) // 16
mask = torch.ones(gy * gx, dtype=torch.uint8, device=dev)
dgr.set_bin_persistent("sort")
dgr.set_speculative_sort(False)
pl, rg, D = dgr.bin_gaussians(m2, depths, radii, co, mask, W, H)
torch.cuda.synchronize()
buf = list(dgr._SORT_SCRATCH.values())[0]
nd = ((D + 1) * 4 + 255) // 256 * 256
kB = buf[2 * nd:2 * nd + 4 * D].view(torch.int32).cpu().long() & 0xFFFFFFFF
vB = buf[3 * nd:3 * nd + 4 * D].view(torch.int32).cpu().long() & 0xFFFFFFFF
for s in list(range(0, 8)) + list(range(150, 166)) + list(range(370, 400)):
    print(f"slot {s}: nwin {kB[s] >> 24} lo {(kB[s] >> 16) & 255} wend {kB[s] & 0xFFFF} s_off[lo] {vB[s] >> 16} g0 {vB[s] & 0xFFFF}")
