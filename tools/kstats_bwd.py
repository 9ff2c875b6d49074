"""debug: work decomposition of K10 (build a variant with -DGSR_STATS and point GSRASTER_LIB at it)"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch
import diff_gaussian_rasterization as dgr
import synthetic_scene as S
dev = torch.device("cuda:0")
W, H, N = 1920, 1080, 1_000_000
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = S.make_gaussians(N, W, H, seed=0, device=dev)
cam = S.orbit_cameras(8, W, H, device=dev)[view]
rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
rast = dgr.GaussianRasterizer(rs)
lib = ctypes.CDLL(dgr._lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
gg = {k: v.requires_grad_(True) for k, v in g.items()}
m2, rgb, co, radii, depths = rast.preprocess_gaussians(gg["means3D"], gg["scales"], gg["rotations"], gg["shs"], gg["opacities"], {})
img, D, _, nc = rast.render_gaussians(m2, co, rgb, depths, radii, None, None, {})
torch.cuda.synchronize(); lib.gsr_debug_stats(buf, 1)
(img * torch.rand(3, H, W, device=dev)).sum().backward()
torch.cuda.synchronize(); lib.gsr_debug_stats(buf, 1)
v = list(buf)
print(f"D={D} tiles=8160 quadrant-waves={4*8160}")
print(f"chunk-waves walked          : {v[6]}  ({v[6]/(4*8160):.1f} per wave)")
print(f"entries loaded (lane slots) : {v[0]}")
print(f"candidates after quad cull  : {v[1]}  ({100*v[1]/max(v[0],1):.1f} % of loaded)")
print(f"candidates evaluated        : {v[2]}")
print(f"taken slots (>=1 lane)      : {v[3]}  ({100*v[3]/max(v[2],1):.1f} % of evaluated)")
print(f"taking lanes                : {v[4]}  ({v[4]/max(v[3],1):.1f} lanes per taken slot)")
print(f"reduction batches           : {v[5]}  ({v[3]/max(v[5],1):.2f} taken slots per batch of 7)")
