"""debug: occupancy statistics of the composite walk (build with GSR_DEFINES=-DGSR_STATS)"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch
import diff_gaussian_rasterization as dgr
import synthetic_scene as S
dev = torch.device("cuda:0")
W, H, N = 1920, 1080, 1_000_000
g = S.make_gaussians(N, W, H, seed=0, device=dev)
cam = S.orbit_cameras(8, W, H, device=dev)[0]
rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
rast = dgr.GaussianRasterizer(rs)
lib = ctypes.CDLL(dgr._lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    m2, rgb, co, radii, depths = rast.preprocess_gaussians(g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    torch.cuda.synchronize(); lib.gsr_debug_stats(buf, 1)
    img, D, _, nc = rast.render_gaussians(m2, co, rgb, depths, radii, None, None, {})
    torch.cuda.synchronize(); lib.gsr_debug_stats(buf, 1)
v = list(buf)
print(f"D={D}  waves={4 * 8160}")
print(f"entries loaded by waves (lane-slots): {v[0]}  -> {v[0] / (4 * 8160):.0f} per wave")
print(f"  relevant after quadrant test:       {v[1]}  ({100 * v[1] / max(v[0], 1):.1f} % of loaded)")
print(f"  walked entries with >=1 taking lane: {v[2]}  ({100 * v[2] / max(v[1], 1):.1f} % of walked... walked=stat2 denominators differ if early exit)")
print(f"  taking lanes total:                  {v[3]}  -> {v[3] / max(v[2], 1):.1f} lanes per useful entry")
