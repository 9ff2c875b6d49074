"""bench.py -- the hot path's headline measurement on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (N > 1, one rank per GPU)

A STEP is one training iteration of the hot path on one batch of synthetic cameras, through the
reference's own call sequence (train_internal.py:134-329): start_strategy_final -> GT staging ->
distributed_preprocess3dgs_and_all2all_final (activations + K1 per camera + the sparse exchange)
-> render_final (K3-K8) -> batched_loss_computation (band-local L1 + SSIM) -> backward (K10, mirror
exchange, K11, activation backward) -> finish_strategy_final -> grad /= bsz -> Adam step -> zero_grad.
Nothing is skipped or cached inside the timed region.

Workload (BASELINE.json configs[1], SURVEY.md §8(d) "C2"): a synthetic scene of 1,000,000 Gaussians
("Mip360-bicycle sized"), 1920x1080 cameras, SH degree 3, fp32.  At N GPUs the scene is sharded over
the ranks (contiguous shards) and the batch holds N cameras (bsz = N, Grendel's batched pixel
partition), so per-GPU work is constant: "weak" scaling, value = images / second of the whole job.
Inputs (parameters, cameras, uint8 ground-truth images) are resident in HBM before the timed region.

One JSON line is printed by rank 0; besides the contract fields it carries
  roofline     : HBM roofline of the dominant HIP kernel, from HIP events recorded inside the timed
                 steps on the kernels' stream (algorithmic bytes per launch are SURVEY.md §8(d)'s
                 formulas, restated in DESIGN.md);
  cpu_baseline : oracle/gsraster_ref.c (the plain-C "port") timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "grendel-gs_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes(kernel, N, P, D, Px, tiles, sh_coeffs=16):
    """SURVEY.md §8(d) per-launch algorithmic HBM bytes (stated again in DESIGN.md)"""
    in_per_g = 44 + 12 * sh_coeffs  # xyz 12 + scale 12 + rot 16 + opacity 4 + SH
    if kernel == "preprocess_forward":
        return N * (in_per_g + 44)
    if kernel == "preprocess_backward":
        return N * (in_per_g + 44 + 36 + in_per_g)
    if kernel == "binning":
        key_bits = max(1, math.ceil(math.log2(max(tiles, 2))))
        passes = math.ceil(key_bits / 8)
        return P * (8 + 4 + 4) + P * 4 * 16 + D * 12 + D * 16 * passes + D * 8
    if kernel == "composite_forward":
        return 40 * D + 20 * Px
    if kernel == "composite_backward":
        return 76 * D + 20 * Px
    return 0


def cpu_baseline(W, H, n_total, seconds=12.0):
    """the C restatement (oracle/, "port") on a 1/16-area crop x 1/16 of the Gaussians, all host cores"""
    from oracle import cref as C
    import synthetic_scene as S

    w, h, n = W // 4, H // 4, n_total // 16
    g = S.make_gaussians(n, w, h, seed=0)
    cam = S.orbit_cameras(8, w, h)[0]
    kw = dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              W=w, H=h, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=3)
    keys = ["means3D", "scales", "rotations", "shs", "opacities"]
    mask = torch.ones((h + 15) // 16, (w + 15) // 16, dtype=torch.bool)
    bg = torch.zeros(3)
    wgt = torch.rand(3, h, w, generator=torch.Generator().manual_seed(1))

    def one():
        m2, rgb, co, radii, depths, cov3D, clamped = C.preprocess_forward(*[g[k] for k in keys], **kw)
        pl, ranges, _ = C.bin_and_sort(m2, radii, depths, mask, w, h)
        img, fT, nc = C.render_forward(m2, co, rgb, mask, bg, w, h, pl, ranges)
        d2, dco, drgb = C.render_backward(m2, co, rgb, mask, bg, w, h, pl, ranges, fT, nc, wgt)
        C.preprocess_backward(g["means3D"], g["scales"], g["rotations"], g["shs"], radii, cov3D, clamped, d2, dco,
                              drgb, **kw)

    one()
    t0 = time.time()
    it = 0
    while time.time() - t0 < seconds or it < 3:
        one()
        it += 1
    dt = (time.time() - t0) / it
    return {"value": 1.0 / dt, "unit": "images/s", "cores": C.num_threads(), "kind": "port",
            "sample": f"{n} Gaussians, {w}x{h} (1/16 of the Gaussians on a 1/16-area image), rasterizer "
                      f"fwd+bwd only (no loss/optimizer), {it} iterations, oracle/gsraster_ref.c with OpenMP"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views", type=int, default=8, help="distinct synthetic cameras to cycle through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--render-steps", type=int, default=20, help="forward-only views/sec leg (untimed by driver)")
    a = ap.parse_args()

    import synthetic_scene as S
    import utils.general_utils as utils

    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    utils.init_distributed(backend="nccl" if world > 1 else None)
    rank = utils.GLOBAL_RANK
    bsz = max(1, world)
    utils.set_args(utils.default_args(bsz=bsz))
    utils.set_img_size(a.height, a.width)
    utils.set_cur_iter(1)

    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    W, H = a.width, a.height
    model = S.SyntheticGaussianModel(a.gaussians, W, H, seed=0, rank=rank, world_size=world, device=dev)
    n_views = max(a.views, bsz)
    cameras = S.orbit_cameras(n_views, W, H, device=dev)
    for k, cam in enumerate(cameras):
        cam.original_image_backup = S.make_gt_image(W, H, seed=1 + k, device=dev)  # preloaded to HBM
    history = DivisionStrategyHistoryFinal(S.SyntheticDataset(cameras), world, rank)
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    pipe = type("Pipe", (), {"debug": False})()
    from fused_optim import FusedAdam

    opt = FusedAdam(model.param_groups(), lr=0.0, eps=1e-15)  # scene/gaussian_model.py:292 settings

    state = {"it": 0}

    def batch():
        s = (state["it"] * bsz) % n_views
        state["it"] += 1
        return [cameras[(s + j) % n_views] for j in range(bsz)]

    def train_step():
        cams = batch()
        utils.set_cur_iter(utils.get_cur_iter() + bsz)
        strategies, tasks = start_strategy_final(cams, history)
        load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies,
                                                           mode="train")
        images, masks = render_final(pkg, strategies)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
        loss.backward()
        finish_strategy_final(cams, history, strategies, stats)
        opt.step(grad_scale=1.0 / bsz)  # grad /= bsz (train_internal.py:319-324) folded into the update
        opt.zero_grad(set_to_none=True)
        for cam in cams:
            cam.original_image = None

    def render_step():
        with torch.no_grad():
            cams = batch()
            strategies, tasks = start_strategy_final(cams, history)
            pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies,
                                                               mode="test")
            images, _ = render_final(pkg, strategies)
        return images

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    for _ in range(a.warmup):
        train_step()
    dgr.kernel_timer.reset()
    dgr.kernel_timer.enabled = True
    dt = timed(train_step, a.steps)
    dgr.kernel_timer.enabled = False
    torch.cuda.synchronize()
    ksum = dgr.kernel_timer.summary_ms()
    D = int(getattr(dgr._RenderGaussians, "last_num_rendered", 0) or 0)

    # forward-only leg: rendered views / second (second half of BASELINE.json's metric)
    for _ in range(3):
        render_step()
    dt_r = timed(render_step, a.render_steps)

    if rank != 0:
        return
    value = bsz * a.steps / dt
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    N_local = model._xyz.shape[0]
    Px = W * H  # at bsz = world every rank renders about one full image per step
    P_render = N_local if world == 1 else None
    kern = {}
    for name, (cnt, ms) in ksum.items():
        nbytes = algorithmic_bytes(name, N_local, P_render or N_local, D, Px, tiles)
        kern[name] = {"launches": cnt, "avg_ms": round(ms, 4), "algo_MB": round(nbytes / 1e6, 2),
                      "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
    hip = {k: v for k, v in kern.items()}
    dom = max(hip, key=lambda k: hip[k]["avg_ms"] * hip[k]["launches"]) if hip else None
    roofline = None
    if dom:
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ach = hip[dom]["GBps"]
        roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None, "traffic": traffic,
                    "avg_ms": hip[dom]["avg_ms"], "algorithmic_bytes": int(hip[dom]["algo_MB"] * 1e6),
                    "note": "composite kernels are VALU/exp-bound by construction (DESIGN.md); the HBM "
                            "fraction is reported because north_star fixes HBM as the yardstick"}
    out = {
        "metric": "training iters/sec (fwd+bwd)",
        "value": round(value, 3),
        "unit": "images/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"synthetic Mip360-bicycle-sized scene: {a.gaussians} Gaussians total, SH degree 3, "
                               f"{W}x{H} cameras, full train iteration (activations, preprocess, exchange, render, "
                               f"L1+SSIM loss, backward, Adam)",
                   "gaussians_total": a.gaussians, "gaussians_per_gpu": N_local, "image": [W, H],
                   "bsz": bsz, "parallelism": f"pixel-partition x{world}, Gaussian-sharded x{world}",
                   "num_rendered_pairs_D": D, "seed": 0},
        "rendered_views_per_sec": round(bsz * a.render_steps / dt_r, 3),
        "kernels": kern,
        "roofline": roofline,
        "reference_published": {"a100_bicycle_1gpu_images_per_s": 16.6, "note": "README.md:342 of the reference; "
                                "other hardware, real data at 1237x822 -- not comparable, hence vs_baseline null"},
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, H, a.gaussians)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
