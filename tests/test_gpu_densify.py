"""N4: row primitives (gsr_group_rows / gsr_gather_rows) and the densification mirror (densification_ops.py)
against the boolean-indexing restatement of the reference's methods (oracle/densify_oracle.py) -- bit-exact:
rows are moved, not recomputed, and torch.normal draws the same samples on the same device."""
import copy

import pytest
import torch

import densification_ops as D
import diff_gaussian_rasterization as dgr
import synthetic_scene as S
from fused_optim import FusedAdam
from oracle import densify_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,groups", [(1, 1), (5, 3), (4097, 1), (100000, 8), (300001, 255)])
def test_group_rows_is_stable_partition(device, n, groups):
    g = torch.Generator().manual_seed(n)
    dest = torch.randint(-2, groups + 2, (n,), generator=g, dtype=torch.int32)
    order, counts = dgr.group_rows(dest.to(device), groups)
    ref_order, ref_counts = O.group_rows(dest, groups)
    assert counts == ref_counts
    assert torch.equal(order.cpu(), ref_order)


def test_gather_rows_dense_and_record_matrix(device):
    g = torch.Generator().manual_seed(0)
    n = 50000
    srcs = [torch.rand(n, 3, generator=g), torch.rand(n, 15, 3, generator=g), torch.rand(n, generator=g),
            torch.randint(0, 99, (n, 4), generator=g, dtype=torch.int32)]
    dest = torch.randint(-1, 2, (n,), generator=g, dtype=torch.int32)
    dsrcs = [s.to(device) for s in srcs]
    order, counts = dgr.group_rows(dest.to(device), 2)
    sel = order[:counts[0]].cpu().long()
    outs = dgr.gather_rows(order, counts[0], dsrcs)
    for o, s in zip(outs, srcs):
        assert torch.equal(o.cpu(), s[sel])
    # second group through row0; identity order; packing into / unpacking from one record matrix
    outs1 = dgr.gather_rows(order, counts[1], dsrcs, None, row0=counts[0])
    sel1 = order[counts[0]:counts[0] + counts[1]].cpu().long()
    assert torch.equal(outs1[1].cpu(), srcs[1][sel1])
    widths = [3, 45, 1, 4]
    rec = torch.zeros(counts[0], sum(widths), device=device)
    cols, c = [], 0
    for w in widths:
        cols.append(rec[:, c:c + w])
        c += w
    flat = [s.view(torch.float32).reshape(n, -1) if s.dtype != torch.float32 else s.reshape(n, -1) for s in dsrcs]
    dgr.gather_rows(order, counts[0], flat, cols)
    back = dgr.gather_rows(None, counts[0], cols)
    assert torch.equal(back[1].cpu().reshape(-1, 15, 3), srcs[1][sel])
    assert torch.equal(back[3].view(torch.int32).cpu(), srcs[3][sel])
    with pytest.raises(RuntimeError, match="no CPU"):
        dgr.gather_rows(None, 4, [srcs[0]])


def _model(device, n=30000, seed=0, world=1):
    torch.manual_seed(seed)
    m = S.SyntheticGaussianModel(n, 320, 240, seed=seed, device=device, scale_coef=0.02)
    m.optimizer = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    for _ in range(2):  # populate exp_avg / exp_avg_sq
        for p in m.parameters():
            p.grad = torch.randn_like(p)
        m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    m.percent_dense = 0.01
    m.xyz_gradient_accum = torch.rand(n, 1, device=device) * 0.001
    m.denom = torch.randint(0, 4, (n, 1), device=device).float()  # zeros -> NaN grads -> 0 (reference rule)
    m.max_radii2D = torch.zeros(n, device=device)
    m.sum_visible_count_in_one_batch = torch.rand(n, device=device)
    m.send_to_gpui_cnt = torch.randint(0, 9, (n, world), dtype=torch.int, device=device)
    return m


def _state(m):
    out = {}
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        out[g["name"]] = p.detach()
        st = m.optimizer.state[p]
        out[g["name"] + ".exp_avg"] = st["exp_avg"]
        out[g["name"] + ".exp_avg_sq"] = st["exp_avg_sq"]
        assert p is getattr(m, D._ATTR[g["name"]])
    for s in D._STATS:
        out[s] = getattr(m, s)
    return out


def test_densify_and_prune_matches_reference_restatement(device):
    a, b = _model(device), _model(device)
    extent = 4.0
    torch.manual_seed(123)
    D.densify_and_prune(a, 0.0002, 0.05, extent, 20)
    torch.manual_seed(123)
    O.densify_and_prune(b, 0.0002, 0.05, extent, 20)
    sa, sb = _state(a), _state(b)
    assert sa["xyz"].shape[0] != 30000  # something was cloned / split / pruned
    for k in sb:
        assert sa[k].shape == sb[k].shape, k
        assert torch.equal(sa[k], sb[k]), k
    # the optimizer keeps stepping on the new tensors
    for p in a.parameters():
        p.grad = torch.ones_like(p)
    a.optimizer.step()


def test_prune_points_all_and_none(device):
    m = _model(device, n=2000)
    ref = copy.deepcopy(_state(m))
    D.prune_points(m, torch.zeros(2000, dtype=torch.bool, device=device))
    for k, v in _state(m).items():
        assert torch.equal(v, ref[k]), k
    D.prune_points(m, torch.ones(2000, dtype=torch.bool, device=device))
    assert all(v.shape[0] == 0 for v in _state(m).values())


def test_training_with_densification_end_to_end(device):
    """the whole loop the reference runs (train_internal.py:134-329 + densification.py:5-86) on the mirror: iterate,
    accumulate the means2D-gradient statistics, densify_and_prune in between, keep training on the
    re-keyed optimizer.  The scene grows / shrinks, nothing goes non-finite, and the loss keeps falling."""
    import utils.general_utils as utils
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    N, W, H = 15000, 320, 208
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    utils.set_args(utils.default_args(bsz=1))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    cams = S.orbit_cameras(8, W, H, device=device)[:3]
    bg = torch.zeros(3, device=device)
    pipe = type("P", (), {"debug": False})()
    teacher = S.SyntheticGaussianModel(N, W, H, seed=11, device=device, scale_coef=0.01)
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    with torch.no_grad():
        for cam in cams:
            st, _ = start_strategy_final([cam], hist)
            pkg = distributed_preprocess3dgs_and_all2all_final([cam], teacher, pipe, bg, batched_strategies=st,
                                                               mode="test")
            cam.original_image_backup = (render_final(pkg, st)[0][0].clamp(0, 1) * 255).round().to(torch.uint8)
    m = S.SyntheticGaussianModel(N, W, H, seed=11, device=device, scale_coef=0.01)  # the teacher, perturbed
    with torch.no_grad():
        g = torch.Generator().manual_seed(0)
        m._features_dc += 1.0 * torch.randn(m._features_dc.shape, generator=g).to(device)
        m._opacity += 0.5 * torch.randn(m._opacity.shape, generator=g).to(device)
        m._xyz += 0.01 * torch.randn(m._xyz.shape, generator=g).to(device)
    m.optimizer = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    m.percent_dense = 0.01
    dev = device
    m.xyz_gradient_accum = torch.zeros(N, 1, device=dev)
    m.denom = torch.zeros(N, 1, device=dev)
    m.max_radii2D = torch.zeros(N, device=dev)
    m.sum_visible_count_in_one_batch = torch.zeros(N, device=dev)
    m.send_to_gpui_cnt = torch.zeros(N, 1, dtype=torch.int, device=dev)
    sizes, losses = [N], []
    for it in range(60):
        cam = cams[it % 3]
        utils.set_cur_iter(it + 1)
        st, tasks = start_strategy_final([cam], hist)
        load_camera_from_cpu_to_all_gpu([cam], st, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final([cam], m, pipe, bg, batched_strategies=st)
        images, masks = render_final(pkg, st)
        stats = [c["stats_collector"] for c in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, [cam], masks, st, stats)
        loss.backward()
        finish_strategy_final([cam], hist, st, stats)
        losses.append(loss.item())
        with torch.no_grad():  # densification.py:13-25
            vis = pkg["batched_locally_preprocessed_visibility_filter"][0]
            radii = pkg["batched_locally_preprocessed_radii"][0]
            m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis].float())
            D.add_densification_stats(m, pkg["batched_locally_preprocessed_mean2D"][0], vis)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        cam.original_image = None
        if it in (19, 39):
            with torch.no_grad():  # threshold at the 95th percentile of the accumulated statistic: ~5 % densify
                gr = (m.xyz_gradient_accum / m.denom.clamp(min=1)).squeeze(1)
                thr = torch.quantile(gr[m.denom.squeeze(1) > 0], 0.95).item()
                D.densify_and_prune(m, thr, 0.005, 4.0, None)
            sizes.append(m._xyz.shape[0])
            for name in D._STATS:
                assert getattr(m, name).shape[0] == m._xyz.shape[0], name
    assert len(set(sizes)) > 1, sizes                       # the scene was actually re-sized
    assert all(torch.isfinite(p).all() for p in m.parameters())
    assert all(l == l for l in losses)
    assert sum(losses[-6:]) < sum(losses[:6]), (losses[:6], losses[-6:])  # still improving after two re-sizings


def test_densification_statistics_without_boolean_indexing(device):
    """add_densification_stats / update_densification_stats (round 6: no `nonzero` host syncs in the per-iteration
    statistics) against the reference's statements with boolean indexing (densification.py:13-25,
    scene/gaussian_model.py:1046-1052), bit for bit, accumulated over three views"""
    n = 20011
    g = torch.Generator().manual_seed(4)

    class M:
        pass

    a, b, c = M(), M(), M()
    for m in (a, b, c):
        m.xyz_gradient_accum = torch.zeros((n, 1), device=device)
        m.denom = torch.zeros((n, 1), device=device)
        m.max_radii2D = torch.zeros((n,), device=device)
    for view in range(3):
        radii = torch.randint(0, 40, (n,), generator=g).to(torch.int32)
        radii[torch.rand(n, generator=g) < 0.3] = 0
        radii = radii.to(device)
        vis = radii > 0
        rec = torch.randn((n, 9), generator=g).to(device)
        rec[~vis] = 0.0  # K10 never touches the row of an invisible Gaussian
        vp = M()
        vp.grad = rec[:, 0:2]  # a strided view of the [P,9] record, like the op's means2D gradient
        # the reference's statements
        a.max_radii2D[vis] = torch.max(a.max_radii2D[vis], radii[vis].float())
        a.xyz_gradient_accum[vis] += torch.norm(vp.grad[vis, :2], dim=-1, keepdim=True)
        a.denom[vis] += 1
        # this build: the method with the reference's signature, and the fused form the loop uses
        b.max_radii2D = torch.maximum(b.max_radii2D, radii.float())
        D.add_densification_stats(b, vp, vis)
        D.update_densification_stats(c, vp, radii)
    for m in (b, c):
        assert torch.equal(m.max_radii2D, a.max_radii2D)
        assert torch.equal(m.denom, a.denom)
    assert torch.equal(b.xyz_gradient_accum, a.xyz_gradient_accum)  # torch's own norm, no indexing
    # the one-launch HIP kernel (gsr_densify_stats): sqrt(fma(gy, gy, gx * gx)) -- torch's reduction may round the sum of
    # squares differently in the last bit
    dev_ = (c.xyz_gradient_accum - a.xyz_gradient_accum).abs().max().item()
    assert dev_ <= 4e-7 * a.xyz_gradient_accum.abs().max().item(), dev_
    print("max |fused - torch| of the accumulated gradient norms:", dev_)


def test_reset_opacity_matches_reference_rule(device):
    m = _model(device, n=5000)
    before = m.get_opacity.detach().clone()
    D.reset_opacity(m)
    want = torch.min(before, torch.full_like(before, 0.01))
    assert torch.allclose(m.get_opacity, want, rtol=1e-5, atol=1e-7)
    st = m.optimizer.state[m._opacity]
    assert float(st["exp_avg"].abs().sum()) == 0.0 and float(st["exp_avg_sq"].abs().sum()) == 0.0
    assert m.optimizer.param_groups[3]["params"][0] is m._opacity
