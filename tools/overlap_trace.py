"""Evidence for the side-stream exchange (north_star: "... exchange ... overlapped with the backward on a side HIP
stream"): runs training iterations of a B-camera batch through the mirror with the exchange FORCED on (a one-rank RCCL
group on this single GPU: the all-to-all-v is a real device-side RCCL call, every row goes to "rank 0"), once with
the per-camera pipelining on the side stream and once without, and prints step times.  Run it under
    rocprofv3 --kernel-trace -d DIR -- python tools/overlap_trace.py
and feed the database to tools/overlap_summary.py, which reports how long kernels of the exchange (pack / RCCL /
unpack / scatter-add) ran CONCURRENTLY with the composite kernels of other cameras.
Usage (GPU box): python tools/overlap_trace.py [--gaussians N] [--bsz B] [--steps K]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--bsz", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    import gaussian_renderer as gr
    import synthetic_scene as S
    import utils.general_utils as utils
    from fused_optim import FusedAdam
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    W, H, B = a.width, a.height, a.bsz
    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 1
    utils.set_args(utils.default_args(bsz=B))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    model = S.SyntheticGaussianModel(a.gaussians, W, H, seed=0, device=dev, on_device=True)
    cams = S.orbit_cameras(max(B, 2), W, H, device=dev)[:B]
    for k, c in enumerate(cams):
        c.original_image_backup = S.make_gt_image(W, H, seed=1 + k, device=dev)
    bg = torch.zeros(3, device=dev)
    pipe = type("P", (), {"debug": False})()
    opt = FusedAdam(model.param_groups(), lr=0.0, eps=1e-15)

    class ForcedGroup:  # size() == 1 selects the W = 1 shortcut; the mirror must take the exchange path instead
        def __init__(self, g):
            self.g = g

        def size(self):
            return 1

        def rank(self):
            return 0

    def step(hist):
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
        strategies, tasks = start_strategy_final(cams, hist)
        load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
        pkg = gr.distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
        # force the exchange on the batched state (rank 0 "sends" every visible Gaussian to itself over RCCL)
        utils.DEFAULT_GROUP = dist.group.WORLD
        lists = [pkg[f"batched_{n}_redistributed"] for n in ("rgb", "conic_opacity", "radii", "depths")]
        m2, rgb, co, radii, depths, sizes, (events, token), _pending = gr._batched_exchange_final(
            pkg["batched_locally_preprocessed_mean2D"], *lists, pkg["batched_rasterizers"], strategies, speculate=False)
        for name, val in zip(("means2D", "rgb", "conic_opacity", "radii", "depths"), (m2, rgb, co, radii, depths)):
            pkg[f"batched_{name}_redistributed"] = val
        pkg["_exchange_events"] = events
        if token is not None:
            pkg["batched_cuda_args"][-1]["_exchange_token"] = token
        utils.DEFAULT_GROUP = utils.SingleGPUGroup()
        images, masks = gr.render_final(pkg, strategies)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
        loss.backward()
        finish_strategy_final(cams, hist, strategies, stats)
        opt.step(grad_scale=1.0 / B)
        opt.zero_grad(set_to_none=True)
        return float(sizes[0][0][0])

    for overlap in (True, False):
        gr.set_exchange_overlap(overlap)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
        for _ in range(2):
            rows = step(hist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(hist)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"overlap={overlap}: {dt * 1e3:.3f} ms per {B}-camera step ({rows:.0f} rows exchanged per camera)", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
