"""Phase timeline of pair_scatter_kernel (csrc/binning_rows.h) on the bench scene: needs a build with -DGSR_ROWS_TS
(python tools/build_variant.py rowsts -DGSR_ROWS_TS; GSRASTER_LIB=variants/libgsraster_rowsts.so python tools/rows_timeline.py)."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402
from diff_gaussian_rasterization import _lib  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    W, H = 1920, 1080
    dev = torch.device("cuda:0")
    g = S.make_gaussians(N, W, H, seed=0, device=dev)
    cam = S.orbit_cameras(8, W, H, device=dev)[0]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                           1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                           False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    mask = torch.ones(((H + 15) // 16) * ((W + 15) // 16), dtype=torch.uint8, device=dev)
    for _ in range(4):
        pl, rg, D = dgr.bin_gaussians(m2, depths, radii, co, mask, W, H)
    torch.cuda.synchronize()
    lib = _lib.lib
    lib.gsr_debug_rows_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    nt = min((D + 4095) // 4096 + 68, 16384)
    buf = np.zeros(16384 * 8, dtype=np.uint64)
    assert lib.gsr_debug_rows_ts(buf.ctypes.data, 16384 * 8) == 0
    ts = buf.reshape(16384, 8)[:nt].astype(np.float64) / 100.0  # us (100 MHz clock)
    ts = ts[ts[:, 0] > 0]
    t0 = ts[:, 0].min()
    names = ["decode", "count", "publish+scans", "rank", "look-back", "stores"]
    print(f"D={D} tiles with stamps {len(ts)}; kernel span {ts[:, 6].max() - t0:.1f} us")
    for k, n in enumerate(names):
        d = ts[:, k + 1] - ts[:, k]
        print(f"  {n:14s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    life = ts[:, 6] - ts[:, 0]
    print(f"  tile life      median {np.median(life):6.2f}  p90 {np.percentile(life, 90):6.2f}  max {life.max():6.2f} us")
    start = np.sort(ts[:, 0] - t0)
    for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
        print(f"  {int(q * 100):3d} % of the tiles started by {start[int(q * (len(start) - 1))]:7.1f} us")
    lib.gsr_debug_rows_ts2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf2 = np.zeros(16384 * 8, dtype=np.uint64)
    assert lib.gsr_debug_rows_ts2(buf2.ctypes.data, 16384 * 8) == 0
    t2 = buf2.reshape(16384, 8).astype(np.float64) / 100.0
    for title, c0, names2 in (("seg_scatter_kernel", 0, ["decode", "count+rank+look-back+scatter"]),
                              ("seg_scan_kernel", 4, ["load+count+scan", "look-back"])):
        x = t2[t2[:, c0] > 0][:, c0:c0 + 3]
        if not len(x):
            continue
        z = x[:, 0].min()
        print(f"{title}: {len(x)} tiles, span {x.max() - z:.1f} us")
        for k, n in enumerate(names2):
            d = x[:, k + 1] - x[:, k]
            print(f"  {n:30s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
        st = np.sort(x[:, 0] - z)
        print("  tiles started by: " + ", ".join(f"{int(q * 100)} % {st[int(q * (len(st) - 1))]:.1f} us" for q in (0.5, 0.9, 1.0)))


if __name__ == "__main__":
    main()
