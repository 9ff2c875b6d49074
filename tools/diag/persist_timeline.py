"""Per-phase timeline of the persistent binning kernels (csrc/binning_persist.h) on the bench scene.
Usage (GPU box): GSR_BIN_TIMELINE=1 python tools/diag/persist_timeline.py [--gaussians N] [--band LO HI] [--width W --height H]
Every workgroup's thread 0 stamps the 100 MHz clock at its phase boundaries; printed: per phase, the mean / max duration
over the workgroups that ran it and the time from the kernel's first stamp to the last workgroup leaving the phase."""
import argparse
import ctypes
import math
import os
import sys

os.environ["GSR_BIN_TIMELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402

P_NAMES = {0: "start", 1: "T (K3 + count 0)", 2: "barrier", 19: "S1 (tile-count sums)", 20: "barrier", 31: "S2 (offsets)"}
for p in range(4):
    P_NAMES[3 + 4 * p] = f"B{p} (scatter)"
    P_NAMES[4 + 4 * p] = "barrier"
    if p < 3:
        P_NAMES[5 + 4 * p] = f"A{p + 1} (load + count)"
        P_NAMES[6 + 4 * p] = "barrier"
S_NAMES = {0: "start", 1: "E0 (column counts from rects)", 2: "barrier", 3: "E1 (decode + scatter by column)", 4: "barrier",
           5: "R0 (count rows)", 6: "barrier", 7: "R1 (scatter by row)", 8: "barrier", 9: "T (tile ranges)"}


def report(which, names):
    grid = ctypes.c_int(0)
    buf = np.zeros(4096 * 32, dtype=np.uint64)
    rc = dgr.lib.gsr_bin_timeline(which, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(grid))
    assert rc == 0, rc
    G = grid.value
    if G == 0:
        print("  (no persistent launch)")
        return
    t = buf[:G * 32].reshape(G, 32).astype(np.float64) / 100.0  # us
    t0 = t[:, 0][t[:, 0] > 0].min()
    keys = sorted(names)
    print(f"  grid {G} workgroups; starts spread over {t[:, 0].max() - t0:.1f} us")
    prev = keys[0]
    for k in keys[1:]:
        a, b = t[:, prev], t[:, k]
        ok = (a > 0) & (b > 0)
        if ok.any():
            d = b[ok] - a[ok]
            print(f"  {names[k]:36s} mean {d.mean():7.2f}  max {d.max():7.2f} us   last workgroup done at {b[ok].max() - t0:7.1f} us")
        prev = k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--band", type=int, nargs=2, default=None)
    ap.add_argument("--view", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H = a.width, a.height
    g = S.make_gaussians(a.gaussians, W, H, seed=0, device=dev)
    cam = S.orbit_cameras(8, W, H, device=dev)[a.view]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                           1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                           False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.uint8, device=dev)
    if a.band:
        mask[:a.band[0]] = 0
        mask[a.band[1]:] = 0
    mask = mask.view(-1)
    dgr.set_bin_persistent("both")
    dgr.set_speculative_sort(False)
    for it in range(4):
        pl, rg, D = dgr.bin_gaussians(m2, depths, radii, co, mask, W, H)
        torch.cuda.synchronize()
    print(f"P = {m2.shape[0]}, D = {D}, {W}x{H}, band {a.band}")
    print("prepare kernel:")
    report(0, P_NAMES)
    print("sort kernel:")
    report(1, S_NAMES)


if __name__ == "__main__":
    main()
