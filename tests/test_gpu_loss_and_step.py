"""-m gpu: the fused L1+SSIM HIP loss vs its CPU restatement / golden maps, and one full training
iteration through the reference-shaped call sequence (gaussian_renderer mirror) vs the oracle chain."""
import math
import os

import numpy as np
import pytest
import torch

import synthetic_scene as S
from helpers import KEYS, cam_kwargs, rel_err

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))


def test_fused_loss_matches_golden_maps(device):
    """sums of the golden per-pixel maps produced by the reference's pixelwise_* functions"""
    from diff_gaussian_rasterization import fused_l1_ssim_band

    img = torch.from_numpy(GOLD["loss_img"]).to(device)
    gt = torch.from_numpy(GOLD["loss_gt_u8"]).to(device)
    C, H, W = img.shape
    l1, ssim = fused_l1_ssim_band(img, gt, 0, H)
    assert abs(l1.item() - float(GOLD["loss_l1_map"].astype(np.float64).sum())) < 1e-4 * l1.item()
    assert abs(ssim.item() - float(GOLD["loss_ssim_map"].astype(np.float64).sum())) < 1e-4 * abs(ssim.item())


@pytest.mark.parametrize("H,W,y0,y1", [(83, 131, 0, 83), (96, 160, 16, 64), (1080, 1920, 272, 544)])
def test_fused_loss_fwd_bwd_matches_torch_restatement(device, H, W, y0, y1):
    from diff_gaussian_rasterization import fused_l1_ssim_band
    from oracle.loss_oracle import l1_map, ssim_map

    g = torch.Generator().manual_seed(H + W)
    img = torch.rand(3, H, W, generator=g)
    gt = torch.randint(0, 256, (3, y1 - y0, W), generator=g, dtype=torch.uint8)
    # oracle in float64 on the band, zero padding at the band edges
    x = img[:, y0:y1].double().clone().requires_grad_(True)
    y = torch.clamp(gt.double() / 255.0, 0, 1)
    l1o, sso = l1_map(x, y).sum(), ssim_map(x, y).sum()
    (0.8 * l1o / 7.0 - 0.2 * sso / 7.0).backward()
    xi = img.to(device).requires_grad_(True)
    l1, ss = fused_l1_ssim_band(xi, gt.to(device), y0, y1)
    (0.8 * l1 / 7.0 - 0.2 * ss / 7.0).backward()
    assert abs(l1.item() - l1o.item()) < 2e-5 * l1o.item()
    assert abs(ss.item() - sso.item()) < 2e-5 * abs(sso.item())
    assert rel_err(xi.grad[:, y0:y1], x.grad) < 1e-4
    # rows outside the band get exactly zero gradient
    outside = torch.ones(H, dtype=torch.bool); outside[y0:y1] = False
    assert xi.grad[:, outside.to(device)].abs().sum().item() == 0.0


def test_native_gpu_timer_log(device, tmp_path):
    """--zhx_time: the per-stage GPU times of a logged iteration land in <log_folder>/gpu_time_ws=1_rk=0.log in the format
    the reference's analyze_statistic.py:747-805 parses ("it=<n>, ..." header, then "<stage>: <ms> ms" lines with the stage
    names of :1972-1991) -- for the camera-batched mirror path and for the reference-shaped operator calls"""
    import diff_gaussian_rasterization as dgr
    import utils.general_utils as utils
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

    def parse(path):  # the reference's parser, restated
        its = []
        for line in open(path):
            if line.startswith("it="):
                its.append({"iteration": int(line[3:line.find(",")])})
                continue
            parts = line.split(":")
            if len(parts) == 2:
                its[-1][parts[0]] = float(parts[1].strip().split("ms")[0].strip())
        return its

    N, W, H = 4000, 208, 144
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    log_folder = str(tmp_path / "logs")
    utils.set_args(utils.default_args(bsz=2, zhx_time=True, log_interval=1, log_folder=log_folder))
    utils.set_img_size(H, W)
    utils.set_cur_iter(7)
    model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
    cams = S.orbit_cameras(2, W, H, device=device)
    for k, c in enumerate(cams):
        c.original_image_backup = S.make_gt_image(W, H, seed=10 + k)
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    bg = torch.tensor([0.1, 0.2, 0.3], device=device)
    pipe = type("P", (), {"debug": False})()
    try:
        strategies, tasks = start_strategy_final(cams, hist)
        load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
        images, masks = render_final(pkg, strategies)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
        loss.backward()
        path = os.path.join(log_folder, "gpu_time_ws=1_rk=0.log")
        its = parse(path)
        assert len(its) == 2  # one record per camera of the batch
        for rec in its:
            assert set(dgr._ZHX_STAGES) <= set(rec)
            for stage in ("10 preprocess time", "24 updateDistributedStatLocally.updateTileTouched time",
                          "50 SortPairs time", "70 render time", "b10 render time", "b20 preprocess time"):
                assert 0.0 < rec[stage] < 100.0, (stage, rec[stage])
        # nothing is written when the flag is off
        utils.set_args(utils.default_args(bsz=2, zhx_time=False, log_interval=1, log_folder=log_folder))
        size = os.path.getsize(path)
        strategies, tasks = start_strategy_final(cams, hist)
        pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
        images, masks = render_final(pkg, strategies)
        sum(im.sum() for im in images).backward()
        assert os.path.getsize(path) == size
    finally:
        utils.set_args(utils.default_args(bsz=1))


def test_training_iteration_through_mirror_matches_oracle(device):
    """start_strategy_final -> GT staging -> preprocess(+exchange) -> render_final -> batched loss ->
    backward, world size 1, against the C restatement + torch loss restatement"""
    import utils.general_utils as utils
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)
    from oracle import cref as C
    from oracle.loss_oracle import band_loss

    N, W, H = 4000, 208, 144
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    utils.set_args(utils.default_args(bsz=2))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
    cams = S.orbit_cameras(2, W, H, device=device)
    for k, c in enumerate(cams):
        c.original_image_backup = S.make_gt_image(W, H, seed=10 + k)
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    bg = torch.tensor([0.1, 0.2, 0.3], device=device)
    pipe = type("P", (), {"debug": False})()

    strategies, tasks = start_strategy_final(cams, hist)
    assert tasks == [[(0, 0, utils.TILE_Y), (1, 0, utils.TILE_Y)]]
    load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
    pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
    for key in ("batched_locally_preprocessed_mean2D", "batched_locally_preprocessed_visibility_filter",
                "batched_locally_preprocessed_radii", "batched_rasterizers", "batched_cuda_args",
                "batched_means2D_redistributed", "batched_rgb_redistributed", "batched_conic_opacity_redistributed",
                "batched_radii_redistributed", "batched_depths_redistributed", "gpui_to_gpuj_imgk_size"):
        assert key in pkg
    images, masks = render_final(pkg, strategies)
    stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
    loss, parts = batched_loss_computation(images, cams, masks, strategies, stats)
    loss.backward()
    finish_strategy_final(cams, hist, strategies, stats)
    for st in stats:
        for key in ("forward_render_time", "backward_render_time", "forward_loss_time"):
            assert isinstance(st[key], float)
    # densification's input: NDC-scaled means2D gradient of the locally preprocessed Gaussians
    assert pkg["batched_locally_preprocessed_mean2D"][0].grad is not None

    # ---- oracle: same two cameras, summed loss
    g = {"means3D": model.get_xyz, "scales": model.get_scaling, "rotations": model.get_rotation,
         "shs": model.get_features, "opacities": model.get_opacity}
    g = {k: v.detach().cpu() for k, v in g.items()}
    total = 0.0
    d_means = torch.zeros(N, 3, dtype=torch.float64)
    mask = torch.ones(utils.TILE_Y, utils.TILE_X, dtype=torch.bool)
    for k, cam in enumerate(cams):
        camc = S.orbit_cameras(2, W, H)[k]
        kw = cam_kwargs(camc)
        m2, rgb, co, radii, depths, cov3D, clamped = C.preprocess_forward(*[g[x] for x in KEYS], **kw)
        pl, ranges, _ = C.bin_and_sort(m2, radii, depths, mask, W, H)
        img, fT, nc = C.render_forward(m2, co, rgb, mask, bg.cpu(), W, H, pl, ranges)
        x = img.double().clone().requires_grad_(True)
        l, _, _ = band_loss(x, cam.original_image_backup, H, W)
        l.backward()
        total += l.item()
        d2, dco, drgb = C.render_backward(m2, co, rgb, mask, bg.cpu(), W, H, pl, ranges, fT, nc, x.grad.float())
        d_means += C.preprocess_backward(g["means3D"], g["scales"], g["rotations"], g["shs"], radii, cov3D, clamped,
                                         d2, dco, drgb, **kw)[0].double()
    assert abs(loss.item() - total) < 1e-4 * abs(total)
    e = rel_err(model._xyz.grad, d_means)
    print(f"[mirror iteration] d_xyz rel {e:.2e}")
    assert e < 1e-4


def test_fused_adam_matches_torch_adam(device):
    """N3: same trajectory as stock torch.optim.Adam with the reference's settings, incl. grad / bsz"""
    from fused_optim import FusedAdam

    g = torch.Generator().manual_seed(0)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 4)]
    lrs = [0.00016, 0.0025, 0.000125, 0.05, 0.001]
    pa = [torch.randn(s, generator=g).to(device).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    bsz = 4
    for it in range(5):
        for p, q in zip(pa, pb):
            gr = torch.randn(p.shape, generator=g).to(device)
            gr[::3] = 0.0  # invisible Gaussians: zero gradient, moments still decay
            p.grad = gr.clone()
            q.grad = gr.clone() / bsz
        oa.step(grad_scale=1.0 / bsz)
        ob.step()
    for p, q in zip(pa, pb):
        assert rel_err(p, q) < 1e-6
    for p, q in zip(pa, pb):
        assert rel_err(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]) < 1e-6


def test_exchange_path_on_device_single_rank_rccl(device):
    """the W > 1 code path (need mask stacking, size all-gather, nonzero_static, all_to_all_single and
    its autograd mirror) executed on the GPU through RCCL with a one-rank process group, against the
    W == 1 shortcut: same images, same gradients."""
    import torch.distributed as dist

    import gaussian_renderer as gr
    import utils.general_utils as utils
    from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        N, W, H = 5000, 240, 160
        utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
        utils.set_args(utils.default_args(bsz=2))
        utils.set_img_size(H, W)
        utils.set_cur_iter(1)
        cams = S.orbit_cameras(2, W, H, device=device)
        bg = torch.tensor([0.1, 0.2, 0.3], device=device)
        pipe = type("P", (), {"debug": False})()
        wgt = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(k)).to(device) for k in range(2)]

        def run(group):
            utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = group
            model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
            hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
            strategies, tasks = start_strategy_final(cams, hist)
            pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
            images, _ = gr.render_final(pkg, strategies)
            sum((im * w_).sum() for im, w_ in zip(images, wgt)).backward()
            return [im.detach() for im in images], model._xyz.grad.clone(), model._features_rest.grad.clone(), pkg

        class OneRankButDistributed:  # size() > 1 would need more GPUs; the exchange code only needs the group
            def __init__(self, g):
                self.g = g

            def size(self):
                return 1

            def rank(self):
                return 0

        img_a, gx_a, gf_a, _ = run(utils.SingleGPUGroup())
        # force the exchange: call it directly on the preprocessed package of a fresh model
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = dist.group.WORLD
        model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
        strategies, tasks = start_strategy_final(cams, hist)
        utils.DEFAULT_GROUP = utils.SingleGPUGroup()
        pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
        utils.DEFAULT_GROUP = dist.group.WORLD
        params = [[pkg["batched_means2D_redistributed"][k], pkg["batched_rgb_redistributed"][k],
                   pkg["batched_conic_opacity_redistributed"][k], pkg["batched_radii_redistributed"][k],
                   pkg["batched_depths_redistributed"][k]] for k in range(2)]
        m2, rgb, co, radii, depths, sizes = gr.all_to_all_communication_final(
            pkg["batched_rasterizers"], params, pkg["batched_cuda_args"], strategies)
        assert sizes[0][0][0] == int((params[0][3] > 0).sum().item())  # only visible Gaussians travel
        for name, val in zip(("means2D", "rgb", "conic_opacity", "radii", "depths"), (m2, rgb, co, radii, depths)):
            pkg[f"batched_{name}_redistributed"] = val
        images, _ = gr.render_final(pkg, strategies)
        sum((im * w_).sum() for im, w_ in zip(images, wgt)).backward()
        for a, b in zip(img_a, images):
            assert rel_err(b, a) < 1e-6
        assert rel_err(model._xyz.grad, gx_a) < 1e-5
        assert rel_err(model._features_rest.grad, gf_a) < 1e-5

        # the camera-batched exchange (gsr_exchange_need, packed 11-float records, regroup of two cameras) over RCCL
        model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
        utils.DEFAULT_GROUP = utils.SingleGPUGroup()
        pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
        utils.DEFAULT_GROUP = dist.group.WORLD
        lists = [pkg[f"batched_{n}_redistributed"] for n in ("rgb", "conic_opacity", "radii", "depths")]
        m2, rgb, co, radii, depths, sizes, (events, token), _pending = gr._batched_exchange_final(
            pkg["batched_locally_preprocessed_mean2D"], *lists, pkg["batched_rasterizers"], strategies, speculate=False)
        pkg["_exchange_events"] = events
        if token is not None:
            pkg["batched_cuda_args"][-1]["_exchange_token"] = token
        assert sizes[0][0][1] == int((lists[2][1] > 0).sum().item())
        for name, val in zip(("means2D", "rgb", "conic_opacity", "radii", "depths"), (m2, rgb, co, radii, depths)):
            pkg[f"batched_{name}_redistributed"] = val
        images, _ = gr.render_final(pkg, strategies)
        sum((im * w_).sum() for im, w_ in zip(images, wgt)).backward()
        for a, b in zip(img_a, images):
            assert rel_err(b, a) < 1e-6
        assert rel_err(model._xyz.grad, gx_a) < 1e-5
        assert rel_err(model._features_rest.grad, gf_a) < 1e-5
        assert pkg["batched_locally_preprocessed_mean2D"][1].grad is not None  # densification's input survives
    finally:
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
        if created:
            dist.destroy_process_group()


def test_fused_activations_match_getters(device):
    """a19: exp / normalize / sigmoid / cat of GaussianModel's getters, forward and backward"""
    from diff_gaussian_rasterization import fused_activations

    m = S.SyntheticGaussianModel(3001, 320, 200, seed=1, device=device)
    with torch.no_grad():
        m._rotation[5] = 0.0  # degenerate quaternion: normalize's eps path
    ws = [torch.rand(s, generator=torch.Generator().manual_seed(i)).to(device)
          for i, s in enumerate([(3001, 3), (3001, 4), (3001, 1), (3001, 16, 3)])]
    outs = fused_activations(m._scaling, m._rotation, m._opacity, m._features_dc, m._features_rest)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    got = [p.grad.clone() for p in (m._scaling, m._rotation, m._opacity, m._features_dc, m._features_rest)]
    for p in m.parameters():
        p.grad = None
    refs = (m.get_scaling, m.get_rotation, m.get_opacity, m.get_features)
    sum((o * w).sum() for o, w in zip(refs, ws)).backward()
    want = [p.grad for p in (m._scaling, m._rotation, m._opacity, m._features_dc, m._features_rest)]
    for o, r in zip(outs, refs):
        assert rel_err(o, r) < 1e-6
    for g_, w_ in zip(got, want):
        assert rel_err(g_, w_) < 1e-5


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_preprocess_raw_equals_getters_then_op(device, deg):
    """the RAW-parameter entry (activations fused into K1/K11) == getters + reference-shaped op, fwd and bwd"""
    from diff_gaussian_rasterization import GaussianRasterizer
    from helpers import settings_from

    N, W, H = 20000, 320, 208
    cam = S.orbit_cameras(4, W, H, device=device)[1]
    rast = GaussianRasterizer(settings_from(cam, torch.zeros(3), sh_degree=deg))
    gen = torch.Generator().manual_seed(3)
    ws = [torch.rand(s, generator=gen).to(device) for s in [(N, 2), (N, 3), (N, 4)]]

    def run(raw):
        m = S.SyntheticGaussianModel(N, W, H, seed=8, device=device, scale_coef=0.01, sh_degree=deg)
        with torch.no_grad():
            m._rotation *= 1.7  # un-normalised quaternions
        if raw:
            outs = rast.preprocess_gaussians_raw(m._xyz, m._scaling, m._rotation, m._features_dc, m._features_rest,
                                                 m._opacity, {})
        else:
            outs = rast.preprocess_gaussians(m.get_xyz, m.get_scaling, m.get_rotation, m.get_features, m.get_opacity, {})
        (outs[0] * ws[0]).sum().add((outs[1] * ws[1]).sum()).add((outs[2] * ws[2]).sum()).backward()
        return outs, [p.grad for p in (m._xyz, m._scaling, m._rotation, m._features_dc, m._features_rest, m._opacity)]

    oa, ga = run(True)
    ob, gb = run(False)
    assert torch.equal(oa[3], ob[3])
    for a, b in zip(oa[:3], ob[:3]):
        assert rel_err(a, b) < 1e-6
    for a, b in zip(ga, gb):
        assert rel_err(a, b) < 1e-5


def test_short_training_run_reduces_loss(device):
    """end-to-end sanity of the whole iteration (mirror + fused loss + FusedAdam): fit a perturbed copy of a
    scene to ground-truth renders of the original for 40 iterations; the loss must drop substantially"""
    import utils.general_utils as utils
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    N, W, H = 20000, 320, 208
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
            img = render_final(pkg, st)[0][0]
            cam.original_image_backup = (img.clamp(0, 1) * 255).round().to(torch.uint8)
    student = S.SyntheticGaussianModel(N, W, H, seed=11, device=device, scale_coef=0.01)
    with torch.no_grad():
        g = torch.Generator().manual_seed(0)
        student._features_dc += 1.0 * torch.randn(student._features_dc.shape, generator=g).to(device)
        student._opacity += 0.5 * torch.randn(student._opacity.shape, generator=g).to(device)
        student._xyz += 0.01 * torch.randn(student._xyz.shape, generator=g).to(device)
    opt = FusedAdam(student.param_groups(), lr=0.0, eps=1e-15)
    losses = []
    for it in range(40):
        cam = cams[it % len(cams)]
        utils.set_cur_iter(it + 1)
        st, tasks = start_strategy_final([cam], hist)
        load_camera_from_cpu_to_all_gpu([cam], st, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final([cam], student, pipe, bg, batched_strategies=st)
        images, masks = render_final(pkg, st)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, [cam], masks, st, stats)
        loss.backward()
        finish_strategy_final([cam], hist, st, stats)
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(loss.item())
    first, lastv = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert all(math.isfinite(x) for x in losses)
    assert lastv < 0.7 * first, f"loss did not drop: {first:.4f} -> {lastv:.4f}"


@pytest.mark.parametrize("deg", [3, 0, 1, 2])
def test_batched_camera_preprocess_equals_per_camera(device, deg):
    """the one-launch-per-batch K1/K11 == B single-camera calls: outputs per camera and SUMMED gradients"""
    import diff_gaussian_rasterization as dgr
    from helpers import settings_from

    N, W, H, B = 30000, 320, 208, 4
    cams = S.orbit_cameras(B, W, H, device=device)
    rasts = [dgr.GaussianRasterizer(settings_from(c, torch.zeros(3), sh_degree=deg)) for c in cams]
    gen = torch.Generator().manual_seed(1)
    ws = [[torch.rand(s, generator=gen).to(device) for s in [(N, 2), (N, 3), (N, 4)]] for _ in range(B)]

    def model():
        m = S.SyntheticGaussianModel(N, W, H, seed=2, device=device, scale_coef=0.01)
        return m, (m._xyz, m._scaling, m._rotation, m._features_dc, m._features_rest, m._opacity)

    ma, ra = model()
    outs_a = [r.preprocess_gaussians_raw(*ra, cuda_args={}) for r in rasts]
    sum((o[0] * w[0]).sum() + (o[1] * w[1]).sum() + (o[2] * w[2]).sum() for o, w in zip(outs_a, ws)).backward()
    mb, rb = model()
    packed = torch.stack([dgr.pack_camera(r.raster_settings) for r in rasts])
    m2, rgb, co, radii, depths = dgr.preprocess_gaussians_raw_batched(*rb, packed, deg, 1.0, W, H)
    sum((m2[k] * ws[k][0]).sum() + (rgb[k] * ws[k][1]).sum() + (co[k] * ws[k][2]).sum() for k in range(B)).backward()
    for k in range(B):
        assert torch.equal(radii[k], outs_a[k][3])
        assert torch.equal(m2[k], outs_a[k][0]) and torch.equal(depths[k], outs_a[k][4])
        assert rel_err(rgb[k], outs_a[k][1]) < 1e-6 and rel_err(co[k], outs_a[k][2]) < 1e-6
    for pa, pb in zip(ra, rb):
        assert rel_err(pb.grad, pa.grad) < 1e-5


@pytest.mark.parametrize("B,deg,N", [(1, 3, 30011), (1, 1, 4099), (3, 3, 30011), (2, 0, 777)])
def test_fused_backward_step_equals_k11_then_adam(device, B, deg, N):
    """FusedAdam(fuse_backward=True): K11 + Adam as ONE kernel == K11, then gsr_adam_step_multi, BIT FOR BIT --
    parameters and both moments after every one of four steps, for the one-camera and the camera-batched kernel,
    active SH degree below the stored one, a Gaussian count that is no multiple of the workgroup, invisible Gaussians
    (zero gradient: the moments still decay).  The incoming gradients are dense tensors from autograd (deterministic);
    in the fused mode the six parameters never get a `.grad`."""
    import diff_gaussian_rasterization as dgr
    from fused_optim import FusedAdam
    from helpers import settings_from

    W, H = 320, 208
    cams = S.orbit_cameras(8, W, H, device=device)[:B]  # rotated by 0, 45, 90 degrees about (0, 0, 6)
    rasts = [dgr.GaussianRasterizer(settings_from(c, torch.zeros(3), sh_degree=deg)) for c in cams]
    packed = torch.stack([dgr.pack_camera(r.raster_settings) for r in rasts])
    rs0 = rasts[0].raster_settings
    gen = torch.Generator().manual_seed(5)
    ws = [[torch.randn(s, generator=gen).to(device) for s in [(N, 2), (N, 3), (N, 4)]] for _ in range(B)]
    names = ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")

    def run(fuse):
        m = S.SyntheticGaussianModel(N, W, H, seed=2, device=device, scale_coef=0.01)
        with torch.no_grad():
            m._xyz[::7] = torch.tensor([-50.0, 0.0, -44.0], device=device)  # behind all three cameras: zero gradient
        opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15, fuse_backward=fuse, grad_scale=1.0 / B)
        snaps = []
        try:
            for it in range(4):
                raw = [getattr(m, n) for n in names]
                m2, rgb, co, radii, depths = dgr.preprocess_gaussians_raw_batched(
                    *raw, packed, deg, 1.0, W, H, tanfov0=(rs0.tanfovx, rs0.tanfovy))
                assert int((radii[0] > 0).sum()) > N // 4
                assert int((torch.stack(list(radii)) <= 0).all(dim=0).sum()) >= N // 7
                loss = sum((m2[k] * ws[k][0]).sum() + (rgb[k] * ws[k][1]).sum() + (co[k] * ws[k][2]).sum()
                           for k in range(B)) * (1.0 + it)
                loss.backward()
                if fuse:
                    assert all(getattr(m, n).grad is None for n in names)
                else:
                    assert all(getattr(m, n).grad is not None for n in names)
                opt.step()
                opt.zero_grad(set_to_none=True)
                snaps.append([getattr(m, n).detach().clone() for n in names] +
                             [opt.state[getattr(m, n)][k].clone() for n in names for k in ("exp_avg", "exp_avg_sq")] +
                             [opt.state[getattr(m, n)]["step"].clone() for n in names])
            assert opt.fused_steps == (4 if fuse else 0) and opt.materialized_steps == 0
        finally:
            opt.set_fuse_backward(False)
        return snaps

    a, b = run(False), run(True)
    for it, (sa, sb) in enumerate(zip(a, b)):
        for j, (x, y) in enumerate(zip(sa, sb)):
            assert torch.equal(x.cpu(), y.cpu()), f"step {it}, tensor {j}: max |diff| {(x - y).abs().max().item():.3e}"
    # ... and the update is not a no-op (with active degree 0 the gradient of _features_rest IS zero and its moments
    # start at zero: that tensor does not move)
    assert not torch.equal(a[0][0], a[3][0]) and (deg == 0 or not torch.equal(a[0][4], a[3][4]))


def test_fused_backward_step_in_the_training_iteration(device):
    """the whole iteration through the mirror with the fused K11 + Adam step: K10's [P,9] record goes into the fused
    kernel through its row stride; the result equals the two-kernel form fed with the SAME record (K10's atomics make
    two runs of the iteration differ in the last bits, so the second optimizer replays the first one's record);
    what happens between backward and step keeps the stock meaning: a replaced parameter is skipped, zero_grad()
    without a step drops the pending gradient, a second backward before the step materializes both."""
    import diff_gaussian_rasterization as dgr
    import utils.general_utils as utils
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

    N, W, H = 20000, 320, 208
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    utils.set_args(utils.default_args(bsz=1))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    cams = S.orbit_cameras(8, W, H, device=device)[:2]
    for k, cam in enumerate(cams):
        cam.original_image_backup = S.make_gt_image(W, H, seed=1 + k, device=device)
    bg = torch.zeros(3, device=device)
    pipe = type("P", (), {"debug": False})()
    names = ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    m = S.SyntheticGaussianModel(N, W, H, seed=3, device=device, scale_coef=0.012)
    twin = S.SyntheticGaussianModel(N, W, H, seed=3, device=device, scale_coef=0.012)
    opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15, fuse_backward=True)
    opt_twin = FusedAdam(twin.param_groups(), lr=0.0, eps=1e-15)

    def backward_only(cam, it):
        utils.set_cur_iter(it + 1)
        st, tasks = start_strategy_final([cam], hist)
        load_camera_from_cpu_to_all_gpu([cam], st, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final([cam], m, pipe, bg, batched_strategies=st)
        images, masks = render_final(pkg, st)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, [cam], masks, st, stats)
        loss.backward()
        return pkg

    try:
        for it in range(3):
            pkg = backward_only(cams[it % 2], it)
            pend = opt._pending
            assert pend is not None and pend.gstride == 9 and all(getattr(m, n).grad is None for n in names)
            assert pkg["batched_locally_preprocessed_mean2D"][0].grad is not None  # densification's statistic: from K10
            # the twin: plain K11 on the SAME record, then the multi-tensor Adam
            grads = dgr.PendingProjectionBackward(
                tuple(getattr(twin, n) for n in names), pend.cams, pend.radii, pend.cov3D, pend.clamped, pend.g_means2D,
                pend.g_conic_opacity, pend.g_rgb, pend.gstride, pend.meta, pend.tanfov0).materialize()
            for n, g in zip(names, grads):
                getattr(twin, n).grad = g.view(getattr(twin, n).shape)
            opt.step()
            opt_twin.step()
            opt.zero_grad(set_to_none=True)
            opt_twin.zero_grad(set_to_none=True)
            for n in names:
                assert torch.equal(getattr(m, n).detach(), getattr(twin, n).detach()), (it, n)
                assert torch.equal(opt.state[getattr(m, n)]["exp_avg_sq"], opt_twin.state[getattr(twin, n)]["exp_avg_sq"])
        assert opt.fused_steps == 3 and opt.materialized_steps == 0

        # zero_grad() without a step drops the pending gradient
        before = [getattr(m, n).detach().clone() for n in names]
        backward_only(cams[0], 3)
        opt.zero_grad(set_to_none=True)
        assert opt._pending is None
        opt.step()
        assert all(torch.equal(getattr(m, n).detach(), b) for n, b in zip(names, before)) and opt.fused_steps == 3

        # a parameter replaced between backward and step (reset_opacity / densification: a NEW tensor in the group) has
        # no gradient and is skipped; the other five step on their materialized gradients
        backward_only(cams[1], 4)
        old_opacity = m._opacity
        new_opacity = torch.nn.Parameter(old_opacity.detach().clone())
        for g in opt.param_groups:
            if g["name"] == "opacity":
                st = opt.state.pop(g["params"][0])
                g["params"][0] = new_opacity
                opt.state[new_opacity] = st
        m._opacity = new_opacity
        kept = new_opacity.detach().clone()
        xyz_before = m._xyz.detach().clone()
        opt.step()
        opt.zero_grad(set_to_none=True)
        assert opt.materialized_steps == 1 and opt.fused_steps == 3
        assert torch.equal(new_opacity.detach(), kept) and not torch.equal(m._xyz.detach(), xyz_before)

        # two backwards before one step: both become ordinary (summed) gradients
        backward_only(cams[0], 5)
        backward_only(cams[1], 6)
        assert opt._pending is None and all(getattr(m, n).grad is not None for n in names)
        assert opt.materialized_steps == 3
        opt.step()
        opt.zero_grad(set_to_none=True)
        # and the next iteration is fused again
        backward_only(cams[0], 7)
        opt.step()
        opt.zero_grad(set_to_none=True)
        assert opt.fused_steps == 4
        assert all(torch.isfinite(getattr(m, n)).all() for n in names)
    finally:
        opt.set_fuse_backward(False)


def test_fused_backward_is_not_offered_a_model_without_sixteen_coefficients(device):
    """the fused K11 + Adam kernel exists for 16-coefficient models (its LDS stage is sized for 45 floats of
    _features_rest per Gaussian); any other model keeps the two-kernel path: ordinary `.grad`s, ordinary step -- and the
    C entry point refuses such a call instead of misreading the rows"""
    import ctypes

    import diff_gaussian_rasterization as dgr
    from fused_optim import FusedAdam
    from helpers import settings_from

    N, W, H = 5000, 240, 160
    m = S.SyntheticGaussianModel(N, W, H, seed=6, device=device, scale_coef=0.012)
    m._features_rest = torch.nn.Parameter(m._features_rest.detach()[:, :8].contiguous())  # 9 coefficients: degree 2
    m.active_sh_degree = 2
    names = ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")
    opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15, fuse_backward=True)
    try:
        cam = S.orbit_cameras(8, W, H, device=device)[0]
        rs = dgr.GaussianRasterizer(settings_from(cam, torch.zeros(3), sh_degree=2)).raster_settings
        packed = dgr.pack_camera(rs).view(1, -1)
        before = m._features_rest.detach().clone()
        m2, rgb, co, radii, depths = dgr.preprocess_gaussians_raw_batched(
            *[getattr(m, n) for n in names], packed, 2, 1.0, W, H, tanfov0=(rs.tanfovx, rs.tanfovy))
        (m2[0].sum() + rgb[0].sum() + co[0].sum()).backward()
        assert opt._pending is None and all(getattr(m, n).grad is not None for n in names)
        opt.step()
        assert opt.fused_steps == 0 and not torch.equal(m._features_rest.detach(), before)
        st = opt.state
        VP, D, I64 = ctypes.c_void_p * 6, ctypes.c_double * 6, ctypes.c_int64 * 6
        rc = dgr.lib.gsr_preprocess_backward_adam_raw_batched(
            N, 1, 2, 9, *[getattr(m, n).data_ptr() for n in names[:2]], 1.0, *[getattr(m, n).data_ptr() for n in names[2:]],
            packed.data_ptr(), W, H, radii[0].data_ptr(), m2[0].data_ptr(), m2[0].data_ptr(), m2[0].data_ptr(),
            co[0].data_ptr(), rgb[0].data_ptr(), 0, VP(*[st[getattr(m, n)]["exp_avg"].data_ptr() for n in names]),
            VP(*[st[getattr(m, n)]["exp_avg_sq"].data_ptr() for n in names]), D(*[0.0] * 6), D(*[0.9] * 6), D(*[0.999] * 6),
            D(*[1e-15] * 6), I64(*[2] * 6), 1.0, None, None)
        assert rc != 0 and b"invalid" in dgr.lib.gsr_error_string(rc).lower()
    finally:
        opt.set_fuse_backward(False)


def test_legacy_render_equals_render_final(device):
    """`gaussian_renderer.render()` (the legacy single-camera surface north_star names) == render_final, W = 1"""
    import gaussian_renderer as gr
    import utils.general_utils as utils
    from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

    N, W, H = 5000, 240, 160
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    utils.set_args(utils.default_args(bsz=1))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    cams = S.orbit_cameras(2, W, H, device=device)[:1]
    bg = torch.tensor([0.1, 0.2, 0.3], device=device)
    pipe = type("P", (), {"debug": False})()
    wgt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)).to(device)
    grads = []
    images = []
    for legacy in (False, True):
        model = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
        strategies, _ = start_strategy_final(cams, hist)
        if legacy:
            pkg = gr.preprocess3dgs_and_all2all(cams, model, pipe, bg, strategies, "train")
            img, mask = gr.render(pkg, strategies[0])
            assert mask.shape == (utils.TILE_Y, utils.TILE_X) and "batched_locally_preprocessed_visibility_filter" in pkg
        else:
            pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
            img = gr.render_final(pkg, strategies)[0][0]
        (img * wgt).sum().backward()
        images.append(img.detach())
        grads.append(model._xyz.grad.clone())
    assert torch.equal(images[0], images[1])
    assert rel_err(grads[1], grads[0]) < 1e-6


def test_speculative_exchange_reads_nothing_back(device):
    """the read-back-free exchange (capacity slabs): in the steady state the whole iteration -- K1, count, size
    all-gather, pack, all-to-all-v over RCCL, unpack, K3-K8, loss, K10, mirror all-to-all-v, K11 -- issues NO
    .cpu() / .item() / .tolist() on a device tensor and no torch.cuda.synchronize(); the only host waits are the
    pair-count poll of the render (inside the C-ABI) and the verification's look at an event that has completed.
    Results equal the world-size-1 path (a one-rank RCCL group: every visible row is sent to the rank itself)."""
    import torch.distributed as dist

    import gaussian_renderer as gr
    import utils.general_utils as utils
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    counts = {"cpu": 0, "item": 0, "tolist": 0, "synchronize": 0, "event_sync": 0}
    saved = (torch.Tensor.cpu, torch.Tensor.item, torch.Tensor.tolist, torch.cuda.synchronize,
             torch.cuda.Event.synchronize)

    def counting(name, fn):
        def f(self, *a, **k):
            if getattr(self, "is_cuda", False):
                counts[name] += 1
            return fn(self, *a, **k)
        return f

    try:
        N, W, H, B = 6000, 240, 160, 2
        utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
        utils.set_args(utils.default_args(bsz=B))
        utils.set_img_size(H, W)
        utils.set_cur_iter(1)
        cams = S.orbit_cameras(B, W, H, device=device)
        for k, c in enumerate(cams):
            c.original_image_backup = S.make_gt_image(W, H, seed=20 + k, device=device)
        bg = torch.tensor([0.1, 0.2, 0.3], device=device)
        pipe = type("P", (), {"debug": False})()

        def iteration(model, hist):
            strategies, tasks = start_strategy_final(cams, hist)
            load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
            pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
            images, masks = gr.render_final(pkg, strategies)
            stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
            loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
            loss.backward()
            finish_strategy_final(cams, hist, strategies, stats)
            return loss, images, pkg

        def fresh():
            m = S.SyntheticGaussianModel(N, W, H, seed=4, device=device, scale_coef=0.012)
            return m, DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)

        model, hist = fresh()
        loss_a, images_a, _ = iteration(model, hist)
        ga = {n: getattr(model, n).grad.clone() for n in ("_xyz", "_features_rest", "_scaling", "_opacity")}

        gr.set_exchange_forced(True)
        gr._PLANNERS.clear()
        before = dict(gr.exchange_stats)
        for overlap in (True, False):
            gr.set_exchange_overlap(overlap)
            gr._PLANNERS.clear()  # no capacities yet: the first iteration below is sized exactly
            model, hist = fresh()
            iteration(model, hist)   # exact sizes (nothing known yet) -> the capacities of the next iterations
            for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
                getattr(model, n).grad = None
            iteration(model, hist)   # speculative; allocator warm
            for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
                getattr(model, n).grad = None
            torch.cuda.synchronize()
            for k in counts:
                counts[k] = 0
            torch.Tensor.cpu, torch.Tensor.item = counting("cpu", saved[0]), counting("item", saved[1])
            torch.Tensor.tolist = counting("tolist", saved[2])

            def sync(*a, **k):
                counts["synchronize"] += 1
                return saved[3](*a, **k)

            def ev_sync(self):
                counts["event_sync"] += 1
                return saved[4](self)

            torch.cuda.synchronize, torch.cuda.Event.synchronize = sync, ev_sync
            try:
                loss_b, images_b, pkg = iteration(model, hist)
            finally:
                (torch.Tensor.cpu, torch.Tensor.item, torch.Tensor.tolist, torch.cuda.synchronize,
                 torch.cuda.Event.synchronize) = saved
            assert counts["cpu"] == counts["item"] == counts["tolist"] == counts["synchronize"] == 0, (overlap, counts)
            assert counts["event_sync"] <= 1, counts  # the verification's look at the (completed) copy of the counts
            torch.cuda.synchronize()
            sizes = pkg["gpui_to_gpuj_imgk_size"]
            assert [sizes[0][0][k] for k in range(B)] == [int((r > 0).sum()) for r in
                                                          pkg["batched_locally_preprocessed_radii"]]
            # padded rows: received tensors are capacity-sized, valid rows first
            for k in range(B):
                rad = pkg["batched_radii_redistributed"][k]
                assert rad.shape[0] % 256 == 0 and int((rad > 0).sum()) == sizes[0][0][k]
            assert abs(loss_b.item() - loss_a.item()) <= 1e-6 * abs(loss_a.item())
            for a_, b_ in zip(images_a, images_b):
                assert rel_err(b_, a_) < 1e-6
            for n, g in ga.items():
                assert rel_err(getattr(model, n).grad, g) < 1e-5, n
        after = gr.exchange_stats
        assert after["speculative"] - before["speculative"] == 4 and after["sized"] - before["sized"] == 2
        assert after["redone"] == before["redone"]
    finally:
        gr.set_exchange_forced(False)
        gr.set_exchange_overlap(True)
        gr._PLANNERS.clear()
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
        if created:
            dist.destroy_process_group()
