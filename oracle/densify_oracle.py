"""TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference ships no tests or golden vectors for this path).

Plain-torch restatement of the reference's densification methods, row selection by boolean indexing exactly as
scene/gaussian_model.py does it (prune_points :835-852 / _prune_optimizer :793-818, cat_tensors_to_optimizer
:854-893, densification_postfix :895-921, densify_and_split :922-970, densify_and_clone :972-1005,
densify_and_prune :1007-1044, all2all_gaussian_state :1073-1098 as a single-process permutation).  Works on any
device; the parity tests run it on the same device as the product so that torch.normal draws the same samples.
Also the torch stand-ins of the two row primitives for the CPU (gloo) tests of the host logic."""
import torch
from torch import nn

ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "scaling": "_scaling", "rotation": "_rotation"}


def group_rows(dest, num_groups):
    d = dest.to(torch.int64)
    key = torch.where((d >= 0) & (d < num_groups), d, torch.full_like(d, num_groups))
    order = torch.sort(key, stable=True).indices.to(torch.int32)
    counts = torch.bincount(key, minlength=num_groups + 1).tolist()
    return order, counts


def gather_rows(order, n_out, srcs, dsts=None, row0=0):
    if dsts is None:
        dsts = [torch.empty((n_out,) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device) for s in srcs]
    for s, d in zip(srcs, dsts):
        rows = s[:n_out] if order is None else s[order[row0:row0 + n_out].long()]
        d[:n_out].copy_(rows.reshape((n_out,) + tuple(d.shape[1:])))
    return dsts


def _prune_optimizer(m, mask):
    out = {}
    for group in m.optimizer.param_groups:
        st = m.optimizer.state.get(group["params"][0], None)
        if st is not None:
            st["exp_avg"] = st["exp_avg"][mask]
            st["exp_avg_sq"] = st["exp_avg_sq"][mask]
            del m.optimizer.state[group["params"][0]]
            group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
            m.optimizer.state[group["params"][0]] = st
        else:
            group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
        out[group["name"]] = group["params"][0]
    return out


def prune_points(m, mask):
    valid = ~mask
    t = _prune_optimizer(m, valid)
    for name, attr in ATTR.items():
        setattr(m, attr, t[name])
    m.xyz_gradient_accum = m.xyz_gradient_accum[valid]
    m.send_to_gpui_cnt = m.send_to_gpui_cnt[valid]
    m.denom = m.denom[valid]
    m.max_radii2D = m.max_radii2D[valid]
    m.sum_visible_count_in_one_batch = m.sum_visible_count_in_one_batch[valid]


def densification_postfix(m, new_xyz, new_f_dc, new_f_rest, new_opacity, new_scaling, new_rotation, new_cnt):
    d = {"xyz": new_xyz, "f_dc": new_f_dc, "f_rest": new_f_rest, "opacity": new_opacity, "scaling": new_scaling,
         "rotation": new_rotation}
    out = {}
    for group in m.optimizer.param_groups:
        ext = d[group["name"]]
        st = m.optimizer.state.get(group["params"][0], None)
        if st is not None:
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del m.optimizer.state[group["params"][0]]
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            m.optimizer.state[group["params"][0]] = st
        else:
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
        out[group["name"]] = group["params"][0]
    for name, attr in ATTR.items():
        setattr(m, attr, out[name])
    n, dev = m._xyz.shape[0], m._xyz.device
    m.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    m.denom = torch.zeros((n, 1), device=dev)
    m.max_radii2D = torch.zeros((n,), device=dev)
    m.sum_visible_count_in_one_batch = torch.zeros((n,), device=dev)
    m.send_to_gpui_cnt = torch.cat((m.send_to_gpui_cnt, new_cnt), dim=0)


def build_rotation(r):
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    rr, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - rr * z)
    R[:, 0, 2] = 2 * (x * z + rr * y)
    R[:, 1, 0] = 2 * (x * y + rr * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - rr * x)
    R[:, 2, 0] = 2 * (x * z - rr * y)
    R[:, 2, 1] = 2 * (y * z + rr * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_split(m, grads, grad_threshold, scene_extent, N=2):
    n_init = m.get_xyz.shape[0]
    padded = torch.zeros((n_init,), device=m._xyz.device)
    padded[: grads.shape[0]] = grads.squeeze()
    sel = torch.where(padded >= grad_threshold, True, False)
    sel = torch.logical_and(sel, torch.max(m.get_scaling, dim=1).values > m.percent_dense * scene_extent)
    stds = m.get_scaling[sel].repeat(N, 1)
    means = torch.zeros((stds.size(0), 3), device=stds.device)
    samples = torch.normal(mean=means, std=stds)
    rots = build_rotation(m._rotation[sel]).repeat(N, 1, 1)
    new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + m.get_xyz[sel].repeat(N, 1)
    new_scaling = torch.log(m.get_scaling[sel].repeat(N, 1) / (0.8 * N))
    densification_postfix(m, new_xyz, m._features_dc[sel].repeat(N, 1, 1), m._features_rest[sel].repeat(N, 1, 1),
                          m._opacity[sel].repeat(N, 1), new_scaling, m._rotation[sel].repeat(N, 1),
                          m.send_to_gpui_cnt[sel].repeat(N, 1))
    prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=sel.device, dtype=bool)))
    prune_points(m, prune_filter)


def densify_and_clone(m, grads, grad_threshold, scene_extent):
    sel = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
    sel = torch.logical_and(sel, torch.max(m.get_scaling, dim=1).values <= m.percent_dense * scene_extent)
    densification_postfix(m, m._xyz[sel], m._features_dc[sel], m._features_rest[sel], m._opacity[sel],
                          m._scaling[sel], m._rotation[sel], m.send_to_gpui_cnt[sel])


def densify_and_prune(m, max_grad, min_opacity, extent, max_screen_size):
    grads = m.xyz_gradient_accum / m.denom
    grads[grads.isnan()] = 0.0
    densify_and_clone(m, grads, max_grad, extent)
    densify_and_split(m, grads, max_grad, extent)
    prune_mask = (m.get_opacity < min_opacity).squeeze()
    if max_screen_size:
        big_vs = m.max_radii2D > max_screen_size
        big_ws = m.get_scaling.max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
    prune_points(m, prune_mask)
