"""CPU oracle (PyTorch, autograd-differentiable) of the Grendel-GS rasterizer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
path (``grendel-gs_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED.  The arithmetic of this path lives in the un-vendored submodule
``submodules/diff-gaussian-rasterization`` (nyu-systems fork, ``.gitmodules:4-6`` of the
reference; no pinned SHA, directory empty in /root/reference), and the reference ships no
tests or golden vectors.  This file restates the published 3DGS
rasterizer algorithm as Grendel splits it into a *preprocess* op and a *render* op
(SURVEY.md Appendix A), anchored on the reference's call sites:

  * preprocess call / output order  gaussian_renderer/__init__.py:949-960
  * render call                      gaussian_renderer/__init__.py:1271-1282
  * partition test                   gaussian_renderer/workload_division.py:721-744
  * SH basis                         utils/sh_utils.py:26-120      (pinned by tests/golden)
  * quaternion -> R, Sigma = R S S R^T  utils/general_utils.py:416-451 (pinned by tests/golden)
  * camera conventions               scene/cameras.py:84-100, utils/graphics_utils.py:42-76
  * tiles                            utils/general_utils.py:78-93
  * means2D.grad convention          scene/gaussian_model.py:1046-1064

Backward formulas are NOT written here: autograd differentiates the forward, with the two
documented deviations of the CUDA backward reproduced explicitly (straight-through at the
0.99 alpha clamp; means2D gradients carried in NDC-scaled units).  Run in float64 this is the
arbiter for every hand-written backward (C restatement and HIP kernels).
"""
import math

import torch

BLOCK_X = 16
BLOCK_Y = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (
    1.0925484305920792,
    -1.0925484305920792,
    0.31539156525252005,
    -1.0925484305920792,
    0.5462742152960396,
)
SH_C3 = (
    -0.5900435899266435,
    2.890611442640554,
    -0.4570457994644658,
    0.3731763325901154,
    -0.4570457994644658,
    1.445305721320277,
    -0.5900435899266435,
)


class _ScaleGrad(torch.autograd.Function):
    """identity forward, gradient multiplied column-wise by ``s``."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(s)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return g * s, None


def sh_to_rgb_raw(deg, shs, dirs):
    """SH colour before the +0.5 / clamp.  shs [N,16,3] (coefficient-major, RGB innermost,
    scene/gaussian_model.py:122-125), dirs [N,3] unit.  Same polynomial as utils/sh_utils.py:57-120."""
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (
                res
                + SH_C2[0] * xy * shs[:, 4]
                + SH_C2[1] * yz * shs[:, 5]
                + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                + SH_C2[3] * xz * shs[:, 7]
                + SH_C2[4] * (xx - yy) * shs[:, 8]
            )
            if deg > 2:
                res = (
                    res
                    + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9]
                    + SH_C3[1] * xy * z * shs[:, 10]
                    + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11]
                    + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                    + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13]
                    + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                    + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15]
                )
    return res


def quat_to_rotmat(q):
    """(r,x,y,z) -> R, same entries as utils/general_utils.py:416-439 but WITHOUT re-normalising
    (inputs arrive activated, scene/gaussian_model.py:114-115)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
        ],
        dim=1,
    ).view(-1, 3, 3)
    return R


def cov3d_from_scale_rot(scales, rotations, scale_modifier):
    """Sigma = R S S^T R^T (utils/general_utils.py:442-451, scene/gaussian_model.py:35-39);
    returns the 6 upper-triangle values in strip_symmetric order (xx,xy,xz,yy,yz,zz)."""
    R = quat_to_rotmat(rotations)
    L = R * (scales * scale_modifier)[:, None, :]  # R @ diag(s)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def tile_grid(W, H):
    return (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y


def get_rect(xy, radii, W, H):
    """tile rect [min,max) of (pixel centre, radius); C truncation toward zero, clamped to the grid
    (SURVEY.md A.2 step 7)."""
    gx, gy = tile_grid(W, H)
    r = radii.to(xy.dtype)
    minx = torch.trunc((xy[:, 0] - r) / BLOCK_X).to(torch.int64).clamp(0, gx)
    miny = torch.trunc((xy[:, 1] - r) / BLOCK_Y).to(torch.int64).clamp(0, gy)
    maxx = torch.trunc((xy[:, 0] + r + (BLOCK_X - 1)) / BLOCK_X).to(torch.int64).clamp(0, gx)
    maxy = torch.trunc((xy[:, 1] + r + (BLOCK_Y - 1)) / BLOCK_Y).to(torch.int64).clamp(0, gy)
    return minx, miny, maxx, maxy


def _preprocess_core(means3D, scales, rotations, shs, opacities, viewmatrix, projmatrix, campos,
                     W, H, tanfovx, tanfovy, sh_degree, scale_modifier):
    """all Gaussians passed in are assumed in front of the near plane (t.z > 0.2)."""
    dt = means3D.dtype
    N = means3D.shape[0]
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    t = ph @ viewmatrix[:, :3]  # row-vector convention, scene/cameras.py:84-99
    p_hom = ph @ projmatrix
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]

    cov3D = cov3d_from_scale_rot(scales, rotations, scale_modifier)

    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    # Jacobian-only clamp to +-1.3 tanfov.  The CUDA backward treats a clamped t.x / t.y as a
    # constant (gradient multiplier 0) and an unclamped one as t itself, so: where(inside, t, const).
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    tx = torch.where((txtz < -limx) | (txtz > limx), (torch.clamp(txtz, -limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (torch.clamp(tytz, -limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack(
        [focal_x / tz, zero, -(focal_x * tx) / (tz * tz), zero, focal_y / tz, -(focal_y * ty) / (tz * tz)],
        dim=1,
    ).view(N, 2, 3)
    Wc = viewmatrix[:3, :3].t()  # math-convention world->camera rotation
    Sig = torch.stack(
        [cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4], cov3D[:, 2], cov3D[:, 4],
         cov3D[:, 5]],
        dim=1,
    ).view(N, 3, 3)
    M = J @ Wc
    cov2 = M @ Sig @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_ok = det != 0
    det_s = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    xy = torch.stack([((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5, ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    minx, miny, maxx, maxy = get_rect(xy.detach(), radius.detach(), W, H)
    area_ok = (maxx - minx) * (maxy - miny) > 0

    d = means3D - campos[None, :]
    dirs = d / d.norm(dim=1, keepdim=True)
    raw = sh_to_rgb_raw(sh_degree, shs, dirs) + 0.5
    clamped = raw < 0
    rgb = torch.clamp(raw, min=0.0)

    valid = det_ok & area_ok
    conic_opacity = torch.cat([conic, opacities.view(N, 1)], dim=1)
    return xy, rgb, conic_opacity, radius.to(torch.int32), tz, valid, cov3D, clamped


def preprocess(means3D, scales, rotations, shs, opacities, *, viewmatrix, projmatrix, campos, W, H,
               tanfovx, tanfovy, sh_degree, scale_modifier=1.0, return_aux=False):
    """K1 (+ K11 through autograd).  Returns (means2D [N,2], rgb [N,3], conic_opacity [N,4],
    radii int32 [N], depths [N]) in the order of gaussian_renderer/__init__.py:949,960.
    Culled Gaussians: radii 0, zeros elsewhere, no gradient."""
    dt = means3D.dtype
    N = means3D.shape[0]
    viewmatrix = viewmatrix.to(dt)
    projmatrix = projmatrix.to(dt)
    campos = campos.to(dt)
    with torch.no_grad():
        ph = torch.cat([means3D, torch.ones(N, 1, dtype=dt)], dim=1)
        tz_all = (ph @ viewmatrix[:, :3])[:, 2]
        front = tz_all > 0.2
        idx0 = front.nonzero().squeeze(1)
        valid0 = _preprocess_core(means3D[idx0], scales[idx0], rotations[idx0], shs[idx0], opacities[idx0],
                                  viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh_degree,
                                  scale_modifier)[5]
        idx = idx0[valid0]
    xy, rgb, co, radius, tz, valid, cov3D, clamped = _preprocess_core(
        means3D[idx], scales[idx], rotations[idx], shs[idx], opacities[idx], viewmatrix, projmatrix, campos,
        W, H, tanfovx, tanfovy, sh_degree, scale_modifier)
    assert bool(valid.all())
    # NDC-scaled gradient convention: incoming means2D grads are divided by (W/2, H/2)
    xy = _ScaleGrad.apply(xy, torch.tensor([2.0 / W, 2.0 / H], dtype=dt))
    means2D = torch.zeros(N, 2, dtype=dt).index_copy(0, idx, xy)
    rgb_o = torch.zeros(N, 3, dtype=dt).index_copy(0, idx, rgb)
    co_o = torch.zeros(N, 4, dtype=dt).index_copy(0, idx, co)
    radii = torch.zeros(N, dtype=torch.int32).index_copy(0, idx, radius)
    depths = torch.zeros(N, dtype=dt).index_copy(0, idx, tz.detach())
    if return_aux:
        cov_o = torch.zeros(N, 6, dtype=dt).index_copy(0, idx, cov3D.detach())
        cl_o = torch.zeros(N, 3, dtype=torch.bool).index_copy(0, idx, clamped)
        return means2D, rgb_o, co_o, radii, depths, cov_o, cl_o
    return means2D, rgb_o, co_o, radii, depths


def get_local2j_ids_bool(H, W, world_size, means2D, radii, dist_global_strategy):
    """K2: [P, ws] bool, Gaussian i is needed by band j iff its tile rect holds a tile id in
    [div[j], div[j+1]) (gaussian_renderer/workload_division.py:721-744).  div is in flattened
    tile ids and always a multiple of TILE_X (bands are whole tile rows)."""
    gx, _ = tile_grid(W, H)
    minx, miny, maxx, maxy = get_rect(means2D.detach(), radii, W, H)
    nonempty = (radii > 0) & (maxx > minx) & (maxy > miny)
    first = miny * gx + minx  # smallest tile id in rect
    last = (maxy - 1) * gx + (maxx - 1)  # largest tile id in rect
    div = dist_global_strategy.to(torch.int64)
    out = torch.zeros(means2D.shape[0], world_size, dtype=torch.bool)
    for j in range(world_size):
        lo, hi = div[j], div[j + 1]
        # rows of the rect are [miny, maxy); band rows are [lo/gx, hi/gx)
        row_lo = torch.div(lo, gx, rounding_mode="floor")
        row_hi = torch.div(hi + gx - 1, gx, rounding_mode="floor")
        out[:, j] = nonempty & (miny < row_hi) & (maxy > row_lo) & (first < hi) & (last >= lo)
    return out


def bin_and_sort(means2D, radii, depths, compute_locally, W, H):
    """K3-K7.  Returns (point_list int64 [D], ranges int64 [tiles,2], tiles_touched [P]).
    key = tile_id << 32 | float32 bits of depth; stable ascending sort; emission order is
    Gaussian index, then ty, then tx."""
    gx, gy = tile_grid(W, H)
    P = means2D.shape[0]
    minx, miny, maxx, maxy = get_rect(means2D.detach(), radii, W, H)
    vis = radii > 0
    mask = compute_locally.view(gy, gx)
    tiles_touched = torch.zeros(P, dtype=torch.int64)
    depth_bits = depths.detach().to(torch.float32).view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    w = torch.where(vis, (maxx - minx).clamp(min=0), torch.zeros_like(minx))
    h = torch.where(vis, (maxy - miny).clamp(min=0), torch.zeros_like(miny))
    cnt = w * h
    gid = torch.repeat_interleave(torch.arange(P, dtype=torch.int64), cnt)
    excl = torch.cumsum(cnt, 0) - cnt
    off = torch.arange(gid.numel(), dtype=torch.int64) - excl[gid]
    wg = w[gid].clamp(min=1)
    ty = miny[gid] + torch.div(off, wg, rounding_mode="floor")
    tx = minx[gid] + off % wg
    keep = mask[ty, tx]
    gid, ty, tx = gid[keep], ty[keep], tx[keep]
    tiles_touched.index_add_(0, gid, torch.ones_like(gid))
    tid = ty * gx + tx
    keys = (tid << 32) | depth_bits[gid]
    order = torch.sort(keys, stable=True).indices
    point_list = gid[order]
    tile_of = tid[order]
    tids = torch.arange(gx * gy, dtype=torch.int64)
    start = torch.searchsorted(tile_of, tids, right=False)
    end = torch.searchsorted(tile_of, tids, right=True)
    ranges = torch.stack([start, end], dim=1)
    return point_list, ranges, tiles_touched


def render(means2D, conic_opacity, rgb, depths, radii, compute_locally, *, bg, W, H, binning=None):
    """K3-K8 (+ K10 through autograd).  Returns (image [3,H,W], final_T [H,W], n_contrib int32 [H,W]).
    Pixels of tiles with compute_locally == False stay 0 (gaussian_renderer/loss_distribution.py:1875;
    SUM all-reduce at train_internal.py:466-469)."""
    dt = means2D.dtype
    gx, gy = tile_grid(W, H)
    bg = bg.to(dt)
    if binning is None:
        binning = bin_and_sort(means2D, radii, depths, compute_locally, W, H)
    point_list, ranges, _ = binning
    # outgoing means2D grads are multiplied by (W/2, H/2)
    m2 = _ScaleGrad.apply(means2D, torch.tensor([0.5 * W, 0.5 * H], dtype=dt))
    rows = []
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    mask = compute_locally.view(gy, gx)
    image = torch.zeros(3, H, W, dtype=dt)
    pieces = []
    for ty in range(gy):
        for tx in range(gx):
            if not bool(mask[ty, tx]):
                continue
            y0, x0 = ty * BLOCK_Y, tx * BLOCK_X
            y1, x1 = min(y0 + BLOCK_Y, H), min(x0 + BLOCK_X, W)
            s, e = int(ranges[ty * gx + tx, 0]), int(ranges[ty * gx + tx, 1])
            ids = point_list[s:e]
            L = ids.numel()
            py, px = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
            npx = py.numel()
            if L == 0:
                pieces.append((y0, y1, x0, x1, bg.view(3, 1, 1).expand(3, y1 - y0, x1 - x0)))
                continue
            pxf = px.reshape(-1).to(dt)
            pyf = py.reshape(-1).to(dt)
            xy = m2[ids]
            con = conic_opacity[ids]
            col = rgb[ids]
            dx = xy[:, 0:1] - pxf[None, :]
            dy = xy[:, 1:2] - pyf[None, :]
            power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
            araw = con[:, 3:4] * torch.exp(torch.clamp(power, max=0.0))
            # min(0.99, .) in the forward, identity in the backward
            alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()
            skip = (power > 0) | (alpha < 1.0 / 255.0)
            a_eff = torch.where(skip, torch.zeros_like(alpha), alpha)
            one_m = 1.0 - a_eff
            test_T = torch.cumprod(one_m.detach(), dim=0)
            stopped = test_T < 0.0001
            # first stopping entry per pixel (L if none); everything from it on is excluded
            stop_idx = torch.where(stopped.any(dim=0), stopped.to(torch.int64).argmax(dim=0),
                                   torch.full((npx,), L, dtype=torch.int64))
            ar = torch.arange(L)[:, None]
            incl = (ar < stop_idx[None, :]) & ~skip
            one_m_incl = torch.where(incl, one_m, torch.ones_like(one_m))
            T_incl = torch.cumprod(one_m_incl, dim=0)
            T_before = torch.cat([torch.ones(1, npx, dtype=dt), T_incl[:-1]], dim=0)
            w = torch.where(incl, a_eff * T_before, torch.zeros_like(a_eff))
            C = (w[:, None, :] * col[:, :, None]).sum(dim=0)  # [3, npx]
            Tf = T_incl[-1]
            out = C + Tf[None, :] * bg[:, None]
            pieces.append((y0, y1, x0, x1, out.view(3, y1 - y0, x1 - x0)))
            with torch.no_grad():
                final_T[y0:y1, x0:x1] = Tf.view(y1 - y0, x1 - x0)
                last = torch.where(incl.any(dim=0), (incl.to(torch.int64) * (ar + 1)).max(dim=0).values,
                                   torch.zeros(npx, dtype=torch.int64))
                n_contrib[y0:y1, x0:x1] = last.view(y1 - y0, x1 - x0).to(torch.int32)
    # assemble differentiably
    if pieces:
        canvas = [[None] * gx for _ in range(gy)]
        for (y0, y1, x0, x1, val) in pieces:
            canvas[y0 // BLOCK_Y][x0 // BLOCK_X] = val
        row_tensors = []
        for ty in range(gy):
            y0 = ty * BLOCK_Y
            y1 = min(y0 + BLOCK_Y, H)
            cols = []
            for tx in range(gx):
                x0 = tx * BLOCK_X
                x1 = min(x0 + BLOCK_X, W)
                v = canvas[ty][tx]
                if v is None:
                    v = torch.zeros(3, y1 - y0, x1 - x0, dtype=dt)
                cols.append(v)
            row_tensors.append(torch.cat(cols, dim=2))
        image = torch.cat(row_tensors, dim=1)
    return image, final_T, n_contrib


def make_camera(W, H, fx=None, fy=None, R=None, T=None, znear=0.01, zfar=100.0, dtype=torch.float32):
    """Camera matrices in the reference's conventions (scene/cameras.py:84-100,
    utils/graphics_utils.py:42-76): world_view_transform = W2C^T, projection^T, full = wv @ proj,
    camera_center = inverse(wv)[3,:3]."""
    fx = 0.9 * W if fx is None else fx
    fy = fx if fy is None else fy
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    R = torch.eye(3, dtype=torch.float64) if R is None else R.to(torch.float64)
    T = torch.zeros(3, dtype=torch.float64) if T is None else T.to(torch.float64)
    Rt = torch.zeros(4, 4, dtype=torch.float64)
    Rt[:3, :3] = R.t()
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    wv = Rt.to(torch.float32).t().contiguous()
    tanx, tany = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = tany * znear, tanx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = P.t().contiguous()
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    campos = wv.inverse()[3, :3]
    return dict(W=W, H=H, tanfovx=tanx, tanfovy=tany, FoVx=fovx, FoVy=fovy,
                viewmatrix=wv.to(dtype), projmatrix=full.to(dtype), campos=campos.to(dtype))
