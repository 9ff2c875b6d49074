"""Time K3-K7 (gsr_bin_prepare + gsr_bin_sort) alone on the bench scene, for one or more library builds.
Usage (GPU box): python tools/binbench.py [--lib variants/libgsraster_X.so ...] [--iters K] [--view V]
Ablation builds may produce wrong lists; nothing downstream consumes them here."""
import argparse
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402

import diff_gaussian_rasterization as dgr  # noqa: E402
import synthetic_scene as S  # noqa: E402
from diff_gaussian_rasterization import _lib  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name in ("gsr_bin_prepare_bytes", "gsr_bin_prepare", "gsr_bin_sort_bytes", "gsr_bin_sort", "gsr_set_bin_persistent",
                 "gsr_set_tile_cull", "gsr_set_bin_rowmajor"):
        if name in ("gsr_set_bin_persistent", "gsr_set_tile_cull", "gsr_set_bin_rowmajor") and not hasattr(lib, name):
            continue  # (a library built before ABI 11)
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--band", type=int, nargs=2, default=None, metavar=("ROW_LO", "ROW_HI"),
                    help="tile rows computed locally (default: the whole frame)")
    ap.add_argument("--modes", nargs="*", default=["off", "prepare", "sort", "both"],
                    help="gsr_set_bin_persistent modes to time on the production library (lists compared bitwise)")
    ap.add_argument("--cull", nargs="*", type=int, default=[0], help="gsr_set_tile_cull modes to time (0 off, 1 on)")
    ap.add_argument("--rows", nargs="*", type=int, default=[0, 1],
                    help="gsr_set_bin_rowmajor modes to time (0: the two-pass pipelines, 1: one pass over the pairs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H = a.width, a.height
    g = S.make_gaussians(a.gaussians, W, H, seed=0, device=dev)
    cam = S.orbit_cameras(8, W, H, device=dev)[a.view]
    rs = dgr.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                           torch.zeros(3, device=dev), 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 3, cam.camera_center, False, False)
    with torch.no_grad():
        m2, rgb, co, radii, depths = dgr.GaussianRasterizer(rs).preprocess_gaussians(
            g["means3D"], g["scales"], g["rotations"], g["shs"], g["opacities"], {})
    P = m2.shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.bool, device=dev)
    if a.band:
        mask[:a.band[0]] = False
        mask[a.band[1]:] = False
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    codes = {"off": 0, "prepare": 1, "sort": 2, "both": 3}
    libs = [(f"production[{m}, cull {c}, rows {r}]", _lib.LIB_PATH, (codes[m], c, r)) for c in a.cull for r in a.rows
            for m in a.modes if not (r and m in ("sort", "both"))] + [(os.path.basename(p), p, None) for p in a.lib]
    ref_lists, ref_cull = None, None
    for name, path, mode in libs:
        lib = load(path)
        if mode is not None:
            assert lib.gsr_set_bin_persistent(mode[0]) == 0
            assert lib.gsr_set_tile_cull(mode[1]) == 0
            assert lib.gsr_set_bin_rowmajor(mode[2]) == 0
            if mode[1] != ref_cull:
                ref_lists, ref_cull = None, mode[1]  # (lists are compared between modes of ONE culling setting)
        ranges = torch.empty((gx * gy + 1, 2), dtype=torch.int32, device=dev)
        nb = lib.gsr_bin_prepare_bytes(P, W, H)
        prep = torch.zeros(nb, dtype=torch.uint8, device=dev)  # (zeros: timing probes with wrong lists must still read valid indices)
        D = ctypes.c_int64(0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_prep = t_sort = 0.0
        for it in range(a.iters + 3):
            ev[0].record()
            rc = lib.gsr_bin_prepare(P, W, H, ptr(m2), ptr(depths), ptr(radii), ptr(co), ptr(mask), ptr(prep), nb,
                                     ctypes.byref(D), stream)
            assert rc == 0, rc
            ev[1].record()
            sb = lib.gsr_bin_sort_bytes(P, D.value, W, H)
            if it == 0:
                scratch = torch.zeros(sb, dtype=torch.uint8, device=dev)
                plist = torch.empty(D.value, dtype=torch.int32, device=dev)
            rc = lib.gsr_bin_sort(P, W, H, ptr(mask), ptr(prep), D.value, ptr(scratch), sb, ptr(plist), ptr(ranges),
                                  stream)
            assert rc == 0, rc
            ev[2].record()
            torch.cuda.synchronize()
            if it >= 3:
                t_prep += ev[0].elapsed_time(ev[1])
                t_sort += ev[1].elapsed_time(ev[2])
        if mode is not None:
            cur = (plist[:D.value].clone(), ranges.clone())
            if ref_lists is None:
                ref_lists = cur
            else:
                same = torch.equal(cur[0], ref_lists[0]) and torch.equal(cur[1], ref_lists[1])
                name += " lists==" + ("ok" if same else "DIFFER")
        print(f"{name:40s} D={D.value:9d} prepare {t_prep / a.iters * 1e3:7.1f} us   sort {t_sort / a.iters * 1e3:7.1f} us"
              f"   total {(t_prep + t_sort) / a.iters * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
