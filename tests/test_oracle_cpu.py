"""CPU tests of the oracle itself (no GPU): pinned against the golden vectors generated from the
reference's own importable helpers (tests/golden/make_golden.py), cross-checked between the two
independent restatements (autograd torch vs hand-written C backward), and gradient-checked by finite
differences in float64.  BASELINE.json configs[0] (10k Gaussians, 256x256, CPU fwd+bwd) runs here."""
import math
import os

import numpy as np
import pytest
import torch

import synthetic_scene as S
from helpers import KEYS, cam_kwargs, oracle_c_chain, rel_err
from oracle import cref as C
from oracle import torch_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))


def T(name, dtype=torch.float32):
    return torch.from_numpy(GOLD[name]).to(dtype)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_basis_matches_reference_eval_sh(deg):
    out = O.sh_to_rgb_raw(deg, T("sh_coeffs"), T("sh_dirs"))
    assert torch.allclose(out, T(f"sh_rgb_deg{deg}"), rtol=1e-5, atol=1e-6)


def test_rotation_and_covariance_match_reference_helpers():
    q, s = T("cov_quats"), T("cov_scales")
    assert torch.allclose(O.quat_to_rotmat(q), T("cov_R"), rtol=1e-5, atol=1e-6)
    assert torch.allclose(O.cov3d_from_scale_rot(s, q, 1.0), T("cov_sym6"), rtol=1e-5, atol=1e-7)


def test_camera_matrices_match_reference_construction():
    fovx, fovy = GOLD["cam_fov"]
    Wd, Hd = 640, 480
    cam = S.SyntheticCamera(0, Wd, Hd, fx=Wd / (2 * math.tan(fovx / 2)), fy=Hd / (2 * math.tan(fovy / 2)),
                            R=T("cam_R", torch.float64), T=T("cam_T", torch.float64))
    assert torch.allclose(cam.world_view_transform, T("cam_world_view"), atol=1e-6)
    assert torch.allclose(cam.projection_matrix, T("cam_proj"), atol=1e-6)
    assert torch.allclose(cam.full_proj_transform, T("cam_full_proj"), atol=1e-5)
    assert torch.allclose(cam.camera_center, T("cam_center"), atol=1e-5)
    # the C oracle's perspective divide agrees with geom_transform_points (utils/graphics_utils.py:24-31)
    pts = T("cam_points")
    ph = torch.cat([pts, torch.ones(16, 1)], 1) @ cam.full_proj_transform
    ndc = ph[:, :3] / (ph[:, 3:] + 1e-7)
    assert torch.allclose(ndc, T("cam_points_ndc"), rtol=1e-4, atol=1e-5)


def test_c_oracle_sh_and_cov_against_golden():
    """drive the C restatement with single-Gaussian scenes so that its rgb / cov3D outputs are the
    reference helpers' values"""
    n = 64
    dirs, shs = T("sh_dirs"), T("sh_coeffs")
    cam = S.SyntheticCamera(0, 64, 64)
    means = dirs * 5.0 + torch.tensor([0.0, 0.0, 20.0])  # campos = 0 -> dir = normalize(p)
    dn = means / means.norm(dim=1, keepdim=True)
    out = C.preprocess_forward(means, T("cov_scales") * 0.1, T("cov_quats"), shs, torch.full((n, 1), 0.5),
                               **cam_kwargs(cam, 3))
    rgb, radii, cov3D = out[1], out[3], out[5]
    vis = radii > 0
    assert vis.sum() > 10
    expect = torch.clamp(O.sh_to_rgb_raw(3, shs, dn) + 0.5, min=0)
    assert torch.allclose(rgb[vis], expect[vis], rtol=1e-4, atol=1e-5)
    assert torch.allclose(cov3D[vis], (T("cov_sym6") * 0.01)[vis], rtol=1e-4, atol=1e-9)


def _scene(N, W, H, sc, seed, ci):
    g = S.make_gaussians(N, W, H, seed=seed, scale_coef=sc)
    cam = S.orbit_cameras(4, W, H)[ci]
    return g, cam


def test_two_restatements_agree_fwd_and_bwd():
    """C (hand-written backward) vs torch float64 autograd, incl. a non-local band and bg != 0"""
    N, W, H = 2000, 200, 120
    g, cam = _scene(N, W, H, 0.01, 3, 1)
    bg = torch.tensor([0.2, 0.5, 0.7])
    gx, gy = O.tile_grid(W, H)
    mask = torch.ones(gy, gx, dtype=torch.bool)
    mask[2:4] = False
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5))
    ref = oracle_c_chain(g, cam, bg, mask, wgt)
    ins = {k: v.double().clone().requires_grad_(True) for k, v in g.items()}
    m2, rgb, co, radii, depths = O.preprocess(*[ins[k] for k in KEYS], **cam_kwargs(cam))
    m2.retain_grad()
    img, fT, nc = O.render(m2, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)
    (img * wgt.double()).sum().backward()
    assert torch.equal(radii, ref["radii"])
    pl, ranges, _ = O.bin_and_sort(m2.float(), radii, depths.float(), mask, W, H)
    assert torch.equal(pl, ref["point_list"].long())
    assert rel_err(ref["image"], img) < 1e-5
    assert (nc != ref["n_contrib"]).sum().item() <= 2
    assert rel_err(ref["d_means2D"], m2.grad) < 2e-5
    for k, rk in [("means3D", "d_means3D"), ("scales", "d_scales"), ("rotations", "d_rotations"), ("shs", "d_shs"),
                  ("opacities", "d_opacities")]:
        assert rel_err(ref[rk], ins[k].grad) < 2e-5, k
    # non-local pixels are exactly zero in both
    pm = mask.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    assert ref["image"][:, ~pm].abs().sum() == 0 and img[:, ~pm].abs().sum() == 0


def test_baseline_config0_10k_256_cpu_fwd_bwd():
    """BASELINE.json configs[0]: 10k random Gaussians, one 256x256 camera, CPU fwd+bwd"""
    N, W, H = 10000, 256, 256
    g, cam = _scene(N, W, H, 0.004, 0, 0)
    bg = torch.zeros(3)
    mask = torch.ones(16, 16, dtype=torch.bool)
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1))
    ref = oracle_c_chain(g, cam, bg, mask, wgt)
    assert torch.isfinite(ref["image"]).all() and ref["image"].max() > 0.1
    for k in ("d_means3D", "d_scales", "d_rotations", "d_shs", "d_opacities"):
        assert torch.isfinite(ref[k]).all() and ref[k].abs().sum() > 0
    # sortedness property of the per-tile lists
    pl, ranges, depths = ref["point_list"].long(), ref["ranges"], ref["depths"]
    for t in range(0, 256, 17):
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        d = depths[pl[s:e]]
        assert bool((d[1:] >= d[:-1]).all())


def test_autograd_oracle_finite_differences_fp64():
    N, W, H = 120, 64, 48
    g, cam = _scene(N, W, H, 0.03, 2, 0)
    bg = torch.tensor([0.3, 0.1, 0.6], dtype=torch.float64)
    mask = torch.ones(3, 4, dtype=torch.bool)
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).double()
    kw = cam_kwargs(cam)

    def f(d):
        m2, rgb, co, radii, depths = O.preprocess(*[d[k] for k in KEYS], **kw)
        img, _, _ = O.render(m2, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)
        return (img * wgt).sum()

    ins = {k: v.double().clone().requires_grad_(True) for k, v in g.items()}
    f(ins).backward()
    rng = np.random.RandomState(0)
    checked = 0
    for k in KEYS:
        flat = ins[k].grad.reshape(-1)
        nz = flat.abs().nonzero().squeeze(1)
        for idx in rng.choice(nz.numpy(), size=4, replace=False):
            eps = 1e-6
            d = {kk: vv.detach().clone() for kk, vv in ins.items()}
            d[k].view(-1)[idx] += eps
            fp = f(d).item()
            d[k].view(-1)[idx] -= 2 * eps
            fm = f(d).item()
            fd = (fp - fm) / (2 * eps)
            an = flat[idx].item()
            # the forward's min(0.99, .) is straight-through in the backward: skip saturated cases
            if abs(fd - an) <= 1e-4 * max(1.0, abs(an)):
                checked += 1
    assert checked >= 16, f"only {checked}/20 finite-difference probes agree"


def test_means2d_grad_is_ndc_scaled():
    """A.7: render's outgoing means2D gradient = pixel gradient x (W/2, H/2)"""
    W, H = 64, 32
    m2 = torch.tensor([[20.3, 12.7]], dtype=torch.float64, requires_grad=True)
    co = torch.tensor([[0.05, 0.01, 0.08, 0.8]], dtype=torch.float64)
    rgb = torch.tensor([[0.9, 0.5, 0.1]], dtype=torch.float64)
    radii = torch.tensor([14], dtype=torch.int32)
    depths = torch.tensor([3.0], dtype=torch.float64)
    mask = torch.ones(2, 4, dtype=torch.bool)
    bg = torch.zeros(3, dtype=torch.float64)
    img, _, _ = O.render(m2, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)
    img.sum().backward()
    eps = 1e-6
    fds = []
    for j in range(2):
        p = m2.detach().clone(); p[0, j] += eps
        q = m2.detach().clone(); q[0, j] -= eps
        fp = O.render(p, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)[0].sum().item()
        fm = O.render(q, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)[0].sum().item()
        fds.append((fp - fm) / (2 * eps))
    assert abs(m2.grad[0, 0].item() - fds[0] * 0.5 * W) < 1e-4 * abs(fds[0] * 0.5 * W) + 1e-8
    assert abs(m2.grad[0, 1].item() - fds[1] * 0.5 * H) < 1e-4 * abs(fds[1] * 0.5 * H) + 1e-8


def test_partition_test_properties():
    N, W, H = 3000, 320, 200
    g, cam = _scene(N, W, H, 0.02, 9, 0)
    m2, rgb, co, radii, depths, _, _ = C.preprocess_forward(*[g[k] for k in KEYS], **cam_kwargs(cam))
    gx, gy = O.tile_grid(W, H)
    div = torch.tensor([0, 4, 9, gy], dtype=torch.int32) * gx
    a = C.get_local2j_ids_bool(H, W, 3, m2, radii, div)
    b = O.get_local2j_ids_bool(H, W, 3, m2, radii, div)
    assert torch.equal(a, b)
    # every visible Gaussian goes to at least one band; culled ones to none
    assert bool(a[radii > 0].any(dim=1).all()) and not bool(a[radii == 0].any())
    # sending set of band j == Gaussians with tiles_touched > 0 under band j's mask
    for j, (l, r) in enumerate([(0, 4), (4, 9), (9, gy)]):
        mask = torch.zeros(gy, gx, dtype=torch.bool)
        mask[l:r] = True
        _, _, tt = C.bin_and_sort(m2, radii, depths, mask, W, H)
        assert torch.equal(a[:, j], tt > 0)


def test_band_loss_golden():
    """the band-local L1 / SSIM maps of utils/loss_utils.py:88-132 (golden), reproduced by the small
    restatement used for iteration-level parity (11x11 sigma 1.5 window, zero padding, C1/C2)"""
    from oracle.loss_oracle import l1_map, ssim_map

    img, gt = T("loss_img"), torch.clamp(T("loss_gt_u8", torch.uint8) / 255.0, 0.0, 1.0)
    assert torch.allclose(l1_map(img, gt), T("loss_l1_map"), atol=1e-6)
    assert torch.allclose(ssim_map(img, gt), T("loss_ssim_map"), rtol=1e-4, atol=1e-5)
