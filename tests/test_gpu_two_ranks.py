"""Two (and three) ranks on ONE MI355X: the whole multi-GPU training iteration with the real HIP kernels.

The GPU box has a single device and RCCL refuses two ranks on one device, so the ranks of this test share
cuda:0 and talk over `gloo`; the two collectives of the path (all_to_all_single, all_gather_into_tensor) are
staged through the host BY THE TEST (a wrapper installed below) because gloo's device support is not what is
under test.  Everything else is the product: Gaussian shards, camera-batched K1, K2 destination masks, the fused
11-float exchange record and its autograd mirror, K3-K8 on row bands (sentinel keys for non-local tiles),
band-local fused L1+SSIM, K10/K11, the load-balancer bookkeeping.  Rank 0 then restates the same partitioned
iteration with the C oracle and compares loss, assembled images and the gathered parameter gradients.
"""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _DoneWork:
    """what an already-completed collective returns for async_op=True"""

    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


def _stage_collectives_through_host():
    a2a, agi = dist.all_to_all_single, dist.all_gather_into_tensor

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, **kw):
        if not output.is_cuda:
            return a2a(output, input, output_split_sizes, input_split_sizes, group=group, **kw)
        o = torch.empty(output.shape, dtype=output.dtype)
        a2a(o, input.detach().cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
        output.copy_(o)

    def all_gather_into_tensor(output, input, group=None, **kw):
        if not output.is_cuda:
            return agi(output, input, group=group, **kw)
        o = torch.empty(output.shape, dtype=output.dtype)
        agi(o, input.detach().cpu().contiguous(), group=group)
        output.copy_(o)
        return _DoneWork() if kw.get("async_op") else None

    dist.all_to_all_single = all_to_all_single
    dist.all_gather_into_tensor = all_gather_into_tensor


def _worker(rank, world, port, bsz, q):
    try:
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        import synthetic_scene as S
        import utils.general_utils as utils
        from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
        from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
        from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                         set_balance_timing, start_strategy_final)

        set_balance_timing("exact")  # the history is inspected after ONE step below
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        utils.init_distributed(backend="gloo")
        _stage_collectives_through_host()
        utils.set_args(utils.default_args(bsz=bsz, save_strategy_history=(world == 2)))
        N, W, H = 6000, 208, (144 if world < 4 else 272)  # 9 tile rows (17 for the 4- and 8-rank partitions)
        utils.set_img_size(H, W)
        utils.set_cur_iter(1)
        model = S.SyntheticGaussianModel(N, W, H, seed=4, rank=rank, world_size=world, device=dev, scale_coef=0.012)
        cams = S.orbit_cameras(max(bsz, 2), W, H, device=dev)[:bsz]
        for k, c in enumerate(cams):
            c.original_image_backup = S.make_gt_image(W, H, seed=10 + k)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
        if bsz < world:  # skewed cost => the cut is not in the middle of the image
            hist.accum_heuristic[cams[0].uid][: utils.TILE_Y // 2] = 3.0
        bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
        pipe = type("P", (), {"debug": False})()

        import gaussian_renderer as gr

        # two passes over the same iteration: the first exchange is sized exactly (nothing known yet), the second one
        # packs into the capacity slabs derived from the first -- no read-back of its counts before the all-to-all;
        # the comparisons below are made on the SECOND pass
        for step in range(2):
            for nm in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
                getattr(model, nm).grad = None
            hist.history.clear()
            strategies, tasks = start_strategy_final(cams, hist)
            load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
            pkg = distributed_preprocess3dgs_and_all2all_final(cams, model, pipe, bg, batched_strategies=strategies)
            images, masks = render_final(pkg, strategies)
            stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
            loss, _ = batched_loss_computation(images, cams, masks, strategies, stats)
            loss.backward()
            finish_strategy_final(cams, hist, strategies, stats)
        assert gr.exchange_stats == {"speculative": 1, "sized": 1, "redone": 0}, gr.exchange_stats
        assert pkg["gpui_to_gpuj_imgk_size"][rank][rank][0] >= 0  # the lazy [W][W][B] list resolves to python ints
        if world == 2:
            assert len(hist.history) == 1 and len(hist.history[0]["all_gpu_running_time"]) == world
        else:  # timings have no consumer (heuristics frozen, history not saved): no gather, nothing logged
            assert len(hist.history) == 0
        for st, strat in zip(stats, strategies):
            if rank not in strat.gpu_ids or world != 2:
                continue
            assert isinstance(st["forward_render_time"], float) and isinstance(st["backward_render_time"], float)

        # ---- collect on the host: SUM-assembled images (train_internal.py:466-469), total loss, gradients
        stack = torch.zeros(bsz, 3, H, W)
        for k, img in enumerate(images):
            if img is not None and img.dim() == 3:
                stack[k] = img.detach().cpu()
        dist.all_reduce(stack)
        tot = loss.detach().cpu().double().reshape(1).clone()
        dist.all_reduce(tot)
        chunk = (N + world - 1) // world
        names = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
        gathered = {}
        for nm in names:
            gten = getattr(model, nm).grad
            gten = torch.zeros_like(getattr(model, nm)) if gten is None else gten
            gten = gten.detach().cpu().contiguous()
            lst = [torch.zeros((min((r + 1) * chunk, N) - r * chunk,) + tuple(gten.shape[1:])) for r in range(world)]
            dist.all_gather(lst, gten)
            gathered[nm] = torch.cat(lst, 0)

        if rank == 0:
            from helpers import KEYS, cam_kwargs, rel_err
            from oracle import cref as C
            from oracle.loss_oracle import band_loss

            full = S.SyntheticGaussianModel(N, W, H, seed=4, device="cpu", scale_coef=0.012)
            leaves = {nm: getattr(full, nm) for nm in names}
            act = {"means3D": full.get_xyz, "scales": full.get_scaling, "rotations": full.get_rotation,
                   "shs": full.get_features, "opacities": full.get_opacity}
            g = {k: v.detach() for k, v in act.items()}
            mask = torch.ones(utils.TILE_Y, utils.TILE_X, dtype=torch.bool)
            total = 0.0
            grads_act = {k: torch.zeros_like(v) for k, v in g.items()}
            for k in range(bsz):
                camc = S.orbit_cameras(max(bsz, 2), W, H)[k]
                kw = cam_kwargs(camc)
                m2, rgb, co, radii, depths, cov3D, clamped = C.preprocess_forward(*[g[x] for x in KEYS], **kw)
                pl, ranges, _ = C.bin_and_sort(m2, radii, depths, mask, W, H)
                img, fT, nc = C.render_forward(m2, co, rgb, mask, bg.cpu(), W, H, pl, ranges)
                err = (img - stack[k]).abs().max().item()
                assert err < 2e-5, f"camera {k}: assembled partitioned render differs from the oracle by {err}"
                dimg = torch.zeros(3, H, W, dtype=torch.float64)
                for gpu in range(world):
                    for (kk, l, r) in tasks[gpu]:
                        if kk != k:
                            continue
                        y0, y1 = l * 16, min(r * 16, H)
                        x = img[:, y0:y1, :].double().clone().requires_grad_(True)
                        lb, _, _ = band_loss(x, cams[k].original_image_backup[:, y0:y1, :], H, W)
                        lb.backward()
                        total += lb.item()
                        dimg[:, y0:y1, :] = x.grad
                d2, dco, drgb = C.render_backward(m2, co, rgb, mask, bg.cpu(), W, H, pl, ranges, fT, nc, dimg.float())
                outs = C.preprocess_backward(g["means3D"], g["scales"], g["rotations"], g["shs"], radii, cov3D,
                                             clamped, d2, dco, drgb, **kw)
                for name, o in zip(KEYS, outs):
                    grads_act[name] += o.reshape(grads_act[name].shape)
            torch.autograd.backward([act[k] for k in KEYS], [grads_act[k] for k in KEYS])
            assert abs(tot.item() - total) < 1e-4 * abs(total), (tot.item(), total)
            for nm in names:
                e = rel_err(gathered[nm], leaves[nm].grad)
                print(f"[{world} ranks, bsz {bsz}] {nm}: rel {e:.2e}", flush=True)
                assert e < 1e-4, f"{nm}: gradient through the partitioned iteration differs, rel {e}"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,bsz", [(2, 1), (2, 2), (3, 2), (4, 1), (8, 4)])
def test_partitioned_training_iteration_on_device(device, world, bsz):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bsz, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        results = [q.get(timeout=420) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("world", [2, 3])
def test_redistribute_gaussians_on_device(device, world):
    """N4 redistribution with the real row kernels (gsr_group_rows / gsr_gather_rows), ranks sharing the GPU"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_workers import redistribution_worker

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=redistribution_worker, args=(r, world, port, True, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        results = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("world", [2, 3])
def test_replicated_gradient_sync_on_device(device, world):
    """N2 with the real row kernels: fused sparse sync == dense sum, ranks sharing the GPU"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_workers import grad_sync_worker

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=grad_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        results = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"
