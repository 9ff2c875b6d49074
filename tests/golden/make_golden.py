"""Generate tests/golden/reference_helpers.npz by IMPORTING the reference's own Python helpers.

Run in the build container only (needs /root/reference; the GPU box never reads it):

    python tests/golden/make_golden.py

What can be pinned against the reference itself (the rasterizer's CUDA source is an
empty submodule and the reference has no tests, so these in-repo pure-PyTorch fragments are the only
executable anchors for the oracle):
  * SH basis                 utils/sh_utils.py:57-120   eval_sh
  * quaternion -> R, R@S     utils/general_utils.py:416-451  build_rotation / build_scaling_rotation,
    Sigma packing            utils/general_utils.py:400-413  strip_symmetric
  * camera matrices          utils/graphics_utils.py:42-76 getWorld2View2 / getProjectionMatrix,
    composed exactly as scene/cameras.py:84-100 does
  * band loss terms          utils/loss_utils.py:88-132 pixelwise_l1_with_mask / pixelwise_ssim_with_mask
`build_rotation` & co. allocate with device="cuda"; the container has no GPU, so torch.zeros is
wrapped to drop the device argument while they run (the arithmetic is untouched).
"""
import contextlib
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)

from utils import general_utils as gu  # noqa: E402
from utils import graphics_utils as gr  # noqa: E402
from utils import loss_utils as lu  # noqa: E402
from utils import sh_utils as sh  # noqa: E402


@contextlib.contextmanager
def cpu_zeros():
    orig = torch.zeros

    def zeros(*a, **k):
        k.pop("device", None)
        return orig(*a, **k)

    torch.zeros = zeros
    try:
        yield
    finally:
        torch.zeros = orig


def main():
    g = torch.Generator().manual_seed(1234)
    out = {}
    # --- SH
    n = 64
    shs = torch.randn(n, 16, 3, generator=g)
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = shs.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        # eval_sh wants [..., C, K]; GaussianModel.get_features yields [N, K, 3]
        out[f"sh_rgb_deg{deg}"] = sh.eval_sh(deg, shs.transpose(1, 2), dirs).numpy()
    # --- rotation / covariance
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.7 - 2.0)
    quats = torch.randn(n, 4, generator=g)
    quats = quats / quats.norm(dim=1, keepdim=True)
    with cpu_zeros():
        R = gu.build_rotation(quats)
        L = gu.build_scaling_rotation(1.0 * scales, quats)
        cov = L @ L.transpose(1, 2)
        sym = gu.strip_symmetric(cov)
    out["cov_scales"] = scales.numpy()
    out["cov_quats"] = quats.numpy()
    out["cov_R"] = R.numpy()
    out["cov_sym6"] = sym.numpy()
    # --- camera (scene/cameras.py:84-100)
    ax = torch.randn(3, generator=g)
    ax = ax / ax.norm()
    th = 0.7
    K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rw2c = torch.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)
    Rcam = Rw2c.t().numpy().astype(np.float64)  # Camera.R is the transposed W2C rotation
    T = np.array([0.3, -0.2, 1.5])
    fovx, fovy = 1.1, 0.8
    wv = torch.tensor(gr.getWorld2View2(Rcam, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    proj = gr.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wv.inverse()[3, :3]
    out["cam_R"] = Rcam
    out["cam_T"] = T
    out["cam_fov"] = np.array([fovx, fovy])
    out["cam_world_view"] = wv.numpy()
    out["cam_proj"] = proj.numpy()
    out["cam_full_proj"] = full.numpy()
    out["cam_center"] = center.numpy()
    pts = torch.randn(16, 3, generator=g)
    out["cam_points"] = pts.numpy()
    out["cam_points_ndc"] = gr.geom_transform_points(pts, full).numpy()
    # --- band loss (loss_distribution.py:2536-2585 uses these two on a row band, no halo)
    Hh, Ww = 40, 56
    img = torch.rand(3, Hh, Ww, generator=g)
    gt_u8 = torch.randint(0, 256, (3, Hh, Ww), generator=g, dtype=torch.uint8)
    gt = torch.clamp(gt_u8 / 255.0, 0.0, 1.0)
    mask = torch.ones(Hh, Ww, dtype=torch.bool)
    out["loss_img"] = img.numpy()
    out["loss_gt_u8"] = gt_u8.numpy()
    out["loss_l1_map"] = lu.pixelwise_l1_with_mask(img, gt, mask).numpy()
    out["loss_ssim_map"] = lu.pixelwise_ssim_with_mask(img, gt, mask).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


def write_reference_utils_names():
    """names defined at module level by the reference's utils/general_utils.py (functions, classes, globals): the
    B2 graft keeps that module, so the mirror may only touch these (tests/test_b1_graft_cpu.py)"""
    import ast
    import os

    src = open("/root/reference/utils/general_utils.py").read()
    names = set()
    for node in ast.parse(src).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            names.add(node.name)
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                for n in ast.walk(t):
                    if isinstance(n, ast.Name):
                        names.add(n.id)
        elif isinstance(node, ast.FunctionDef):
            names.add(node.name)
    # globals created by `global X` assignments inside functions (IMG_H, TILE_X, ...)
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Global):
            names.update(node.names)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_utils_names.txt")
    open(out, "w").write("\n".join(sorted(names)) + "\n")
    print("wrote", out, len(names), "names")


if __name__ == "__main__":
    if "--names-only" not in sys.argv:
        main()
    write_reference_utils_names()
