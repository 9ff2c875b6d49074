"""Shared helpers of the parity tests: seeded scenes, camera kwargs, error metrics."""
import math

import torch

import synthetic_scene as S
from oracle import cref as C
from oracle import torch_oracle as O

KEYS = ["means3D", "scales", "rotations", "shs", "opacities"]


def cam_kwargs(cam, sh_degree=3, device="cpu"):
    return dict(viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
                campos=cam.camera_center.to(device), W=cam.image_width, H=cam.image_height,
                tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=sh_degree)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def frac_bad(a, b, rtol=1e-3, atol=1e-5):
    """fraction of elements outside |a-b| <= atol + rtol*|b| (threshold flips show up here)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    bad = (a - b).abs() > (atol + rtol * b.abs())
    return bad.double().mean().item()


def settings_from(cam, bg, sh_degree=3, device="cuda"):
    from diff_gaussian_rasterization import GaussianRasterizationSettings

    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx / 2),
        tanfovy=math.tan(cam.FoVy / 2), bg=bg.to(device), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
        sh_degree=sh_degree, campos=cam.camera_center.to(device), prefiltered=False, debug=False)


def oracle_c_chain(g, cam, bg, mask, wgt, sh_degree=3):
    """full fwd+bwd through the C restatement; returns dict of everything"""
    kw = cam_kwargs(cam, sh_degree)
    W, H = cam.image_width, cam.image_height
    m2, rgb, co, radii, depths, cov3D, clamped = C.preprocess_forward(*[g[k] for k in KEYS], **kw)
    pl, ranges, tt = C.bin_and_sort(m2, radii, depths, mask, W, H)
    img, fT, nc = C.render_forward(m2, co, rgb, mask, bg, W, H, pl, ranges)
    d2, dco, drgb = C.render_backward(m2, co, rgb, mask, bg, W, H, pl, ranges, fT, nc, wgt)
    dm, ds, dr, dsh, do = C.preprocess_backward(g["means3D"], g["scales"], g["rotations"], g["shs"], radii, cov3D,
                                                clamped, d2, dco, drgb, **kw)
    return dict(means2D=m2, rgb=rgb, conic_opacity=co, radii=radii, depths=depths, cov3D=cov3D, clamped=clamped,
                point_list=pl, ranges=ranges, image=img, final_T=fT, n_contrib=nc, d_means2D=d2, d_conic_opacity=dco,
                d_rgb=drgb, d_means3D=dm, d_scales=ds, d_rotations=dr, d_shs=dsh, d_opacities=do)
