"""one small scene, persistent sort only, one call"""
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
print("survived stop_after", os.environ.get("GSR_BIN_STOP_AFTER"), "D", D, flush=True)
if os.environ.get("GSR_BIN_STOP_AFTER") == "7":
    buf = list(dgr._SORT_SCRATCH.values())[0]
    nd = ((D + 1) * 4 + 255) // 256 * 256
    kB = buf[2 * nd:3 * nd].view(torch.int32).cpu()
    dgr.set_bin_persistent("off")
    dgr.release_workspaces()
    pl, rg, D2 = dgr.bin_gaussians(m2, depths, radii, co, mask, W, H)
    rg = rg.cpu()
    percol = torch.zeros(16, dtype=torch.int64)
    for t in range(gx * gy):
        percol[t % gx] += int(rg[t, 1] - rg[t, 0])
    print("expected total per column", percol.tolist())
    for w in range(2):
        print("WG", w, "first", kB[w * 512:w * 512 + 16].tolist())
        print("WG", w, "total", kB[w * 512 + 256:w * 512 + 272].tolist())
if os.environ.get("GSR_BIN_STOP_AFTER") in ("9", "10", "12"):
    buf = list(dgr._SORT_SCRATCH.values())[0]
    nd = ((D + 1) * 4 + 255) // 256 * 256
    kB = buf[2 * nd:2 * nd + 4 * D].view(torch.int32).cpu()
    vB = buf[3 * nd:3 * nd + 4 * D].view(torch.int32).cpu()
    col = kB & 15
    print("kB sorted by column:", bool((col[1:] >= col[:-1]).all()), "col hist", torch.bincount(col, minlength=16).tolist())
    print("vB range", int(vB.min()), int(vB.max()), "kB range", int(kB.min()), int(kB.max()))
    runs = int((vB[1:] != vB[:-1]).sum()) + 1
    print("runs of equal ids", runs, "first 24 keys", kB[:24].tolist(), "ids", vB[:24].tolist())
    bad = ((kB >> 4) > 7) | ((kB & 15) > 12)
    print("slots with impossible keys:", int(bad.sum()), "first at", (bad.nonzero()[:8, 0].tolist()))
    print("keys 376..400", kB[376:400].tolist())
    print("ids  376..400", vB[376:400].tolist())
    bs = bad.nonzero()[:, 0]
    print("bad by (slot % 512) // 64:", torch.bincount((bs % 512) // 64, minlength=8).tolist(), "by WG", torch.bincount(bs // 4096, minlength=2).tolist())
    print("keys 4090..4110", kB[4090:4110].tolist(), "ids", vB[4090:4110].tolist())
