"""Which Python lines launch the small torch kernels of a training step (fills, copies)?  torch.profiler with stacks over
a few eager iterations of the bench's step at the headline shape; prints every aten op that launched a device kernel,
with device time per step and the innermost repo frames.

    python tools/diag/small_kernels.py [n_gaussians] [steps] [--exchange]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    import synthetic_scene as S
    import utils.general_utils as utils
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    argv = [x for x in sys.argv[1:] if not x.startswith("--")]
    N = int(argv[0]) if len(argv) > 0 else 1_000_000
    steps = int(argv[1]) if len(argv) > 1 else 4
    forced = "--exchange" in sys.argv  # the exchange path too: every visible row "sent" to the rank itself over RCCL
    W, H = 1920, 1080
    dev = torch.device("cuda:0")
    utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = 0, 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    if forced:
        import torch.distributed as dist

        import gaussian_renderer as gr

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
        gr.set_exchange_forced(True)
    utils.set_args(utils.default_args(bsz=1))
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    model = S.SyntheticGaussianModel(N, W, H, seed=0, device=dev)
    cams = S.orbit_cameras(4, W, H, device=dev)
    for k, c in enumerate(cams):
        c.original_image_backup = S.make_gt_image(W, H, seed=1 + k, device=dev)
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    bg = torch.zeros(3, device=dev)
    pipe = type("P", (), {"debug": False})()
    opt = FusedAdam(model.param_groups(), lr=0.0, eps=1e-15, fuse_backward=True)

    def step(it):
        batch = [cams[it % len(cams)]]
        utils.set_cur_iter(utils.get_cur_iter() + 1)
        strategies, tasks = start_strategy_final(batch, hist)
        load_camera_from_cpu_to_all_gpu(batch, strategies, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final(batch, model, pipe, bg, batched_strategies=strategies,
                                                           mode="train")
        images, masks = render_final(pkg, strategies)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, _ = batched_loss_computation(images, batch, masks, strategies, stats)
        loss.backward()
        finish_strategy_final(batch, hist, strategies, stats)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for it in range(6):
        step(it)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for it in range(steps):
            step(it)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if dt <= 0 or not e.key.startswith("aten::"):
            continue
        frames = [f for f in (e.stack or []) if "/repo/" in f or "bench.py" in f][:3]
        rows.append((dt / steps, e.count / steps, f"{e.key} {e.input_shapes}", frames))
    rows.sort(reverse=True)
    print(f"aten ops that launched device kernels, {N} Gaussians, per step over {steps} steps:")
    for dt, cnt, key, frames in rows[:40]:
        print(f"  {dt:8.1f} us  x{cnt:4.1f}  {key}")
        for f in frames:
            print(f"              {f.replace(ROOT + '/', '')}")


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    os._exit(0)  # (a process group that ran captured or forced collectives may hang in its destructor)
