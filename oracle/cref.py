"""ctypes binding of oracle/gsraster_ref.c (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see the header of gsraster_ref.c.  Tensors in / out are CPU torch
tensors (float32 / int32 / uint8, contiguous)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgsraster_ref.so")
_SRC = os.path.join(_HERE, "gsraster_ref.c")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.gsref_bin_count.restype = ctypes.c_int64
        _lib.gsref_num_threads.restype = ctypes.c_int
    return _lib


def _p(t):
    assert t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def num_threads():
    return int(lib().gsref_num_threads())


def preprocess_forward(means3D, scales, rotations, shs, opacities, *, viewmatrix, projmatrix, campos, W, H,
                       tanfovx, tanfovy, sh_degree, scale_modifier=1.0):
    P = means3D.shape[0]
    M = shs.shape[1]
    means3D, scales, rotations, shs, opacities = map(_f32, (means3D, scales, rotations, shs, opacities))
    view, proj, cam = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    means2D = torch.empty(P, 2)
    depths = torch.empty(P)
    radii = torch.empty(P, dtype=torch.int32)
    cov3D = torch.empty(P, 6)
    conic_opacity = torch.empty(P, 4)
    rgb = torch.empty(P, 3)
    clamped = torch.empty(P, 3, dtype=torch.uint8)
    lib().gsref_preprocess_forward(
        ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M), _p(means3D), _p(scales),
        ctypes.c_float(scale_modifier), _p(rotations), _p(shs), _p(opacities), _p(view), _p(proj), _p(cam),
        ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(means2D), _p(depths),
        _p(radii), _p(cov3D), _p(conic_opacity), _p(rgb), _p(clamped))
    return means2D, rgb, conic_opacity, radii, depths, cov3D, clamped


def preprocess_backward(means3D, scales, rotations, shs, radii, cov3D, clamped, dL_dmeans2D, dL_dconic_opacity,
                        dL_drgb, *, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree,
                        scale_modifier=1.0):
    P = means3D.shape[0]
    M = shs.shape[1]
    means3D, scales, rotations, shs = map(_f32, (means3D, scales, rotations, shs))
    view, proj, cam = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    g2, gco, grgb = _f32(dL_dmeans2D), _f32(dL_dconic_opacity), _f32(dL_drgb)
    radii = radii.to(torch.int32).contiguous()
    cov3D = _f32(cov3D)
    clamped = clamped.to(torch.uint8).contiguous()
    dm = torch.empty(P, 3)
    ds = torch.empty(P, 3)
    dr = torch.empty(P, 4)
    dsh = torch.empty(P, M, 3)
    do = torch.empty(P, 1)
    lib().gsref_preprocess_backward(
        ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M), _p(means3D), _p(scales),
        ctypes.c_float(scale_modifier), _p(rotations), _p(shs), _p(view), _p(proj), _p(cam), ctypes.c_int(W),
        ctypes.c_int(H), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(radii), _p(cov3D), _p(clamped), _p(g2),
        _p(gco), _p(grgb), _p(dm), _p(ds), _p(dr), _p(dsh), _p(do))
    return dm, ds, dr, dsh, do


def get_local2j_ids_bool(H, W, world_size, means2D, radii, dist_global_strategy):
    P = means2D.shape[0]
    out = torch.empty(P, world_size, dtype=torch.uint8)
    m2 = _f32(means2D)
    radii = radii.to(torch.int32).contiguous()
    div = dist_global_strategy.to(torch.int32).contiguous()
    lib().gsref_get_local2j_ids_bool(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(world_size),
                                     _p(m2), _p(radii), _p(div), _p(out))
    return out.bool()


def bin_and_sort(means2D, radii, depths, compute_locally, W, H):
    P = means2D.shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2, dp = _f32(means2D), _f32(depths)
    radii = radii.to(torch.int32).contiguous()
    cl = compute_locally.to(torch.uint8).contiguous().view(-1)
    tiles_touched = torch.empty(P, dtype=torch.int32)
    D = int(lib().gsref_bin_count(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(m2), _p(radii), _p(cl),
                                  _p(tiles_touched)))
    point_list = torch.empty(max(D, 1), dtype=torch.int32)
    ranges = torch.empty(gx * gy, 2, dtype=torch.int32)
    lib().gsref_bin_sort(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(m2), _p(dp), _p(radii), _p(cl),
                         ctypes.c_int64(D), _p(point_list), _p(ranges))
    return point_list[:D], ranges, tiles_touched


def render_forward(means2D, conic_opacity, rgb, compute_locally, bg, W, H, point_list, ranges):
    m2, co, col, bgf = _f32(means2D), _f32(conic_opacity), _f32(rgb), _f32(bg)
    cl = compute_locally.to(torch.uint8).contiguous().view(-1)
    out = torch.empty(3, H, W)
    final_T = torch.empty(H, W)
    n_contrib = torch.empty(H, W, dtype=torch.int32)
    pl = point_list.contiguous() if point_list.numel() else torch.zeros(1, dtype=torch.int32)
    lib().gsref_render_forward(ctypes.c_int(W), ctypes.c_int(H), _p(ranges), _p(pl), _p(m2), _p(co), _p(col), _p(cl),
                               _p(bgf), _p(out), _p(final_T), _p(n_contrib))
    return out, final_T, n_contrib


def render_backward(means2D, conic_opacity, rgb, compute_locally, bg, W, H, point_list, ranges, final_T, n_contrib,
                    dL_dpixels):
    P = means2D.shape[0]
    m2, co, col, bgf = _f32(means2D), _f32(conic_opacity), _f32(rgb), _f32(bg)
    cl = compute_locally.to(torch.uint8).contiguous().view(-1)
    g = _f32(dL_dpixels)
    d2 = torch.empty(P, 2)
    dco = torch.empty(P, 4)
    drgb = torch.empty(P, 3)
    pl = point_list.contiguous() if point_list.numel() else torch.zeros(1, dtype=torch.int32)
    lib().gsref_render_backward(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(ranges), _p(pl), _p(m2), _p(co),
                                _p(col), _p(cl), _p(bgf), _p(final_T.contiguous()), _p(n_contrib.contiguous()), _p(g),
                                _p(d2), _p(dco), _p(drgb))
    return d2, dco, drgb


def knn_mean_dist2(points):
    """brute-force restatement of simple_knn._C.distCUDA2 (scene/gaussian_model.py:163-166 of the reference)"""
    pts = _f32(points)
    out = torch.empty(pts.shape[0], dtype=torch.float32)
    lib().gsref_knn_mean_dist2(ctypes.c_int(pts.shape[0]), _p(pts), _p(out))
    return out
