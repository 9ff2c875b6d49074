"""List segments in the composite backward (include/gsraster.h: gsr_render_forward_seg / gsr_render_backward_seg): K8 leaves
a checkpoint every 256 list entries it really walks, K10 runs the segments of a tile in parallel workgroups starting
from the checkpointed transmittance / colour.  The forward's arithmetic is untouched (image bitwise equal with and
without the workspace); the gradients equal the one-segment kernel's up to fp32 rounding and the C oracle's within the
usual 1e-4; a queue that overflows (more boundaries than checkpoint slots) only leaves long tails unsplit."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def _render_chain(device, g, cam, mask, wgt, segments):
    """K1 -> K3-K8 -> K10 through the operator; -> (image, gradients of means2D / conic_opacity / rgb, None)"""
    import diff_gaussian_rasterization as dgr
    from helpers import KEYS, settings_from

    dgr.set_list_segments("always" if segments else False)
    try:
        bg = torch.tensor([0.1, 0.2, 0.3])
        rast = dgr.GaussianRasterizer(settings_from(cam, bg, device=device))
        with torch.no_grad():
            m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[g[k].to(device) for k in KEYS], {})
        m2, rgb, co = [t.detach().clone().requires_grad_(True) for t in (m2, rgb, co)]
        img, _, _, _ = rast.render_gaussians(m2, co, rgb, depths, radii, mask.to(device), None, {"stats_collector": {}})
        (img * wgt.to(device)).sum().backward()
        torch.cuda.synchronize()
        return img.detach(), m2.grad.clone(), co.grad.clone(), rgb.grad.clone(), None
    finally:
        dgr.set_list_segments(True)


def _scene(n, W, H, seed, scale_coef, op_mean):
    import synthetic_scene as S

    g = S.make_gaussians(n, W, H, seed=seed, scale_coef=scale_coef, opacity_logit_mean=op_mean, opacity_logit_std=0.7)
    cam = S.orbit_cameras(4, W, H)[1]
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed + 1))
    return g, cam, wgt


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("band", [False, True])
def test_segmented_backward_equals_one_segment_and_the_oracle(device, band):
    """deep lists (low opacity: thousands of entries walked per tile) on a small frame, whole image and a thin band"""
    import diff_gaussian_rasterization as dgr
    from helpers import elem_excess, oracle_c_chain

    W, H = 320, 208
    g, cam, wgt = _scene(60_000, W, H, seed=11, scale_coef=0.02, op_mean=-3.5)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.bool)
    if band:
        mask[:] = False
        mask[5:7] = True
    dgr.composite_walked(reset=True)
    img_s, d2_s, dco_s, drgb_s, _ = _render_chain(device, g, cam, mask, wgt, segments=True)
    walked_s = dgr.composite_walked(reset=True)
    img_1, d2_1, dco_1, drgb_1, _ = _render_chain(device, g, cam, mask, wgt, segments=False)
    walked_1 = dgr.composite_walked(reset=True)
    tiles = int(mask.sum())
    assert walked_1[1] / tiles > 3 * 256, f"the scene does not walk deep enough to be cut: {walked_1[1] / tiles:.0f} / tile"
    # the same entries are walked either way (no work is created or lost by the cut), the forward is untouched
    assert walked_s == walked_1, (walked_s, walked_1)
    assert torch.equal(img_s, img_1)
    for name, a, b in (("means2D", d2_s, d2_1), ("conic_opacity", dco_s, dco_1), ("rgb", drgb_s, drgb_1)):
        e = _rel(a, b)
        assert e < 2e-5, f"d{name}: segmented vs one-segment backward differ by {e:.2e}"
    ref = oracle_c_chain(g, cam, torch.tensor([0.1, 0.2, 0.3]), mask, wgt)
    assert _rel(img_s.cpu(), ref["image"]) < 1e-5
    for name, a, key in (("means2D", d2_s, "d_means2D"), ("conic_opacity", dco_s, "d_conic_opacity"), ("rgb", drgb_s, "d_rgb")):
        b = ref[key].reshape(a.shape)
        assert _rel(a.cpu(), b) < 1e-4, (name, _rel(a.cpu(), b))
        assert elem_excess(a.cpu(), b) <= 1.0, (name, elem_excess(a.cpu(), b))


def test_checkpoint_queue_overflow_leaves_tails_unsplit(device):
    """more segment boundaries than checkpoint slots (8192): the first 8192 are cut, the rest of every list is one
    tail segment -- same gradients"""
    import diff_gaussian_rasterization as dgr

    W, H = 1920, 1088
    g, cam, wgt = _scene(400_000, W, H, seed=12, scale_coef=0.012, op_mean=-4.0)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.bool)
    dgr.composite_walked(reset=True)
    img_s, d2_s, dco_s, drgb_s, _ = _render_chain(device, g, cam, mask, wgt, segments=True)
    walked = dgr.composite_walked(reset=True)
    assert walked[1] / 256 > 2 * 8192, f"not enough boundaries to overflow the queue: {walked[1] / 256:.0f}"
    img_1, d2_1, dco_1, drgb_1, _ = _render_chain(device, g, cam, mask, wgt, segments=False)
    assert torch.equal(img_s, img_1)
    for name, a, b in (("means2D", d2_s, d2_1), ("conic_opacity", dco_s, dco_1), ("rgb", drgb_s, drgb_1)):
        e = _rel(a, b)
        assert e < 2e-5, f"d{name}: {e:.2e}"


def test_second_backward_over_one_forward_walks_the_segments_again(device):
    """retain_graph / gradcheck run K10 twice over one forward: the workers' ticket is reset per launch (it
    used to be zeroed by the forward only, and a second backward silently dropped every queued segment); and the image
    the segmented backward reads is saved through autograd, so an in-place edit raises instead of corrupting gradients"""
    import diff_gaussian_rasterization as dgr
    from helpers import KEYS, settings_from

    W, H = 320, 208
    g, cam, wgt = _scene(60_000, W, H, seed=13, scale_coef=0.02, op_mean=-3.5)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    mask = torch.ones(gy, gx, dtype=torch.bool)
    dgr.set_list_segments("always")
    try:
        rast = dgr.GaussianRasterizer(settings_from(cam, torch.tensor([0.1, 0.2, 0.3]), device=device))
        with torch.no_grad():
            m2, rgb, co, radii, depths = rast.preprocess_gaussians(*[g[k].to(device) for k in KEYS], {})
        m2, rgb, co = [t.detach().clone().requires_grad_(True) for t in (m2, rgb, co)]
        img, _, _, _ = rast.render_gaussians(m2, co, rgb, depths, radii, mask.to(device), None, {"stats_collector": {}})
        loss = (img * wgt.to(device)).sum()
        first = torch.autograd.grad(loss, (m2, co, rgb), retain_graph=True)
        second = torch.autograd.grad(loss, (m2, co, rgb), retain_graph=True)
        for a, b in zip(first, second):
            assert _rel(a, b) < 2e-6  # (the order of K10's atomics differs between launches)
        with torch.no_grad():
            img.mul_(0.5)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            torch.autograd.grad(loss, (m2, co, rgb))
    finally:
        dgr.set_list_segments(True)
