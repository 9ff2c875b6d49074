"""The PRODUCT training iteration for W in {2, 4, 8} ranks on ONE MI355X over device-local ASYNCHRONOUS stand-in
collectives (tests/fake_world.py): every fake rank owns its TRUE shard of the scene and renders its true row bands, so
the result is checkable -- and nothing is staged through the host, so a missing `wait_event` / `record_stream` between
the exchange's streams and the renderer's shows up as wrong numbers instead of being hidden by a device round trip
(tests/test_gpu_two_ranks.py stages its gloo collectives through the host).

Reference for every case: the SAME scene on one rank (the W = 1 path of the product), with the loss evaluated band by
band on the partition the W ranks used (the band loss is zero-padded at band edges -- loss_distribution.py:2567-2576 of
the reference -- so the W-rank loss is the sum of band losses, not the full-image loss).  Compared, after three steps
with the learning rates at zero (Adam's moments are then LINEAR in the three steps' gradients, so the comparison is
well-conditioned; the first step's exchange is sized exactly, steps two and three pack into capacity slabs without any
read-back): the summed loss of every step, the assembled image of the last step (bitwise), both Adam moments of all six
parameter tensors shard by shard, the order of collectives the ranks agreed on.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu

NAMES = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
STEPS = 3


def _scene(world):
    # every band needs >= 2 tile rows and >= 10 received Gaussians; 45 tile rows
    return dict(N=120_000, Wd=1280, H=720, views=4, scale_coef=0.006)


def _zero_lr_groups(model):
    groups = model.param_groups()
    for g in groups:
        g["lr"] = 0.0
    return groups


def _run_ranks(dev, world, bsz, fuse, overlap=True, live_balancer=False, group=True):
    """-> per-rank results of STEPS product iterations on the fake world"""
    import gaussian_renderer as gr
    import gaussian_renderer.workload_division as wd
    import synthetic_scene as S
    import utils.general_utils as utils
    from fake_world import FakeWorld
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)

    sc = _scene(world)
    N, Wd, H = sc["N"], sc["Wd"], sc["H"]
    swap = None
    fw = FakeWorld(world, dev, swap=swap)
    gr._PLANNERS.clear()
    gr.set_exchange_overlap(overlap)
    gr.set_exchange_speculation(True)
    gr.set_exchange_grouping(group)

    def rank_main(rank):
        utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = rank, 0, world
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = fw.groups[rank]
        # heuristics live (bsz < W on an image above 1000 x 600) unless frozen: the balancer's timing gather then runs
        # through the (fake) device group every step, with the reference's event waits
        utils.set_args(utils.default_args(bsz=bsz, no_heuristics_update=not live_balancer))
        utils.set_img_size(H, Wd)
        utils.set_cur_iter(1)
        wd._BALANCE["mode"] = "exact"
        for k in gr.exchange_stats:
            gr.exchange_stats[k] = 0
        model = S.SyntheticGaussianModel(N, Wd, H, seed=5, rank=rank, world_size=world, device=dev,
                                         scale_coef=sc["scale_coef"])
        cams = S.orbit_cameras(sc["views"], Wd, H, device=dev)
        for k, c in enumerate(cams):
            c.original_image_backup = S.make_gt_image(Wd, H, seed=20 + k, device=dev)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
        bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
        pipe = type("P", (), {"debug": False})()
        opt = FusedAdam(_zero_lr_groups(model), lr=0.0, eps=1e-15, fuse_backward=fuse, grad_scale=1.0 / bsz)
        losses, partitions, images = [], [], None
        for step in range(STEPS):
            batch = [cams[(step * bsz + j) % len(cams)] for j in range(bsz)]
            utils.set_cur_iter(utils.get_cur_iter() + bsz)
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
            losses.append(loss.detach())
            partitions.append([(list(s.gpu_ids), list(s.division_pos)) for s in strategies])
        out = {"rank": rank, "losses": [float(x) for x in losses], "partitions": partitions,
               "images": [None if (im is None or im.dim() != 3) else im.detach().clone() for im in images],
               "exchange": dict(gr.exchange_stats), "fused_steps": opt.fused_steps,
               "moments": {nm: (opt.state[getattr(model, nm)]["exp_avg"].clone(),
                                opt.state[getattr(model, nm)]["exp_avg_sq"].clone()) for nm in NAMES}}
        opt.set_fuse_backward(False)
        return out

    res = fw.run(rank_main)
    return res, fw


def _run_single(dev, world, bsz, fuse, partitions):
    """the same three steps on ONE rank holding the whole scene; loss = sum of the band losses of `partitions`"""
    import gaussian_renderer as gr
    import synthetic_scene as S
    import utils.general_utils as utils
    from diff_gaussian_rasterization import fused_band_loss
    from fused_optim import FusedAdam
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

    sc = _scene(world)
    N, Wd, H = sc["N"], sc["Wd"], sc["H"]
    utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = 0, 0, 1
    utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
    utils.set_args(utils.default_args(bsz=bsz))
    utils.set_img_size(H, Wd)
    utils.set_cur_iter(1)
    gr._PLANNERS.clear()
    model = S.SyntheticGaussianModel(N, Wd, H, seed=5, device=dev, scale_coef=sc["scale_coef"])
    cams = S.orbit_cameras(sc["views"], Wd, H, device=dev)
    gts = [S.make_gt_image(Wd, H, seed=20 + k, device=dev) for k in range(len(cams))]
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    pipe = type("P", (), {"debug": False})()
    opt = FusedAdam(_zero_lr_groups(model), lr=0.0, eps=1e-15, fuse_backward=fuse, grad_scale=1.0 / bsz)
    losses, images = [], None
    n = float(H * Wd * 3)
    for step in range(STEPS):
        idx = [(step * bsz + j) % len(cams) for j in range(bsz)]
        batch = [cams[i] for i in idx]
        utils.set_cur_iter(utils.get_cur_iter() + bsz)
        strategies, _ = start_strategy_final(batch, hist)
        pkg = distributed_preprocess3dgs_and_all2all_final(batch, model, pipe, bg, batched_strategies=strategies,
                                                           mode="train")
        images, _ = render_final(pkg, strategies)
        loss = None
        for k, (gpu_ids, div) in enumerate(partitions[step]):
            for j in range(len(gpu_ids)):
                y0, y1 = div[j] * 16, min(div[j + 1] * 16, H)
                lb, _, _ = fused_band_loss(images[k], gts[idx[k]][:, y0:y1, :].contiguous(), y0, y1, 0.2, n)
                loss = lb if loss is None else loss + lb
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    out = {"losses": losses, "images": [im.detach().clone() for im in images], "fused_steps": opt.fused_steps,
           "moments": {nm: (opt.state[getattr(model, nm)]["exp_avg"].clone(),
                            opt.state[getattr(model, nm)]["exp_avg_sq"].clone()) for nm in NAMES}}
    opt.set_fuse_backward(False)
    return out


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("world,bsz,fuse,live,group", [
    (2, 1, True, False, True), (4, 1, True, True, True), (8, 1, True, False, True),
    # bsz > 1: every rank renders one camera of the batch -> ONE slab exchange per step (set_exchange_grouping) ...
    (2, 2, True, False, True), (4, 2, False, False, True), (8, 4, True, False, True),
    # ... or one per camera, pipelined on the side stream (what a rank that renders two cameras gets)
    (4, 2, True, False, False), (8, 4, True, False, False)])
def test_async_fake_world_matches_single_rank(device, world, bsz, fuse, live, group):
    import gaussian_renderer as gr

    try:
        res, fw = _run_ranks(device, world, bsz, fuse, live_balancer=live, group=group)
    finally:
        gr.set_exchange_grouping(True)
    partitions = res[0]["partitions"]
    assert all(r["partitions"] == partitions for r in res), "the ranks disagree on the partition"
    ref = _run_single(device, world, bsz, fuse, partitions)
    sc = _scene(world)
    N, H = sc["N"], sc["H"]

    # the exchange really ran read-back-free after the first (exactly sized) step, and nothing had to be repeated
    for r in res:
        assert r["exchange"]["sized"] == 1 and r["exchange"]["speculative"] == STEPS - 1, r["exchange"]
        assert r["exchange"]["redone"] == 0, r["exchange"]
        assert r["fused_steps"] == (STEPS if fuse else 0)
    # every rank issued the same collectives in the same order (checked at every rendezvous); there were some
    # (the first step is sized exactly, one exchange per camera; the speculative ones are ONE exchange when grouped)
    a2a = sum(1 for _, tag in fw.log if tag == "all_to_all_single")  # (one log entry per collective, not per rank)
    assert a2a == 2 * bsz + 2 * (STEPS - 1) * (1 if (group or bsz == 1) else bsz), (a2a, bsz, group)

    # loss: the ranks' band losses add up to the single-rank sum of band losses
    for step in range(STEPS):
        tot = sum(r["losses"][step] for r in res)
        assert abs(tot - ref["losses"][step]) <= 2e-6 * abs(ref["losses"][step]), (step, tot, ref["losses"][step])

    # image of the last step: every rank's band is bitwise the single-rank image's rows, zero elsewhere
    for k, (gpu_ids, div) in enumerate(partitions[-1]):
        for j, g in enumerate(gpu_ids):
            y0, y1 = div[j] * 16, min(div[j + 1] * 16, H)
            band = res[g]["images"][k]
            assert band is not None
            assert torch.equal(band[:, y0:y1], ref["images"][k][:, y0:y1]), f"camera {k} band of rank {g} differs"
            assert float(band[:, :y0].abs().sum()) == 0.0 and float(band[:, y1:].abs().sum()) == 0.0

    # Adam's moments (linear in the three steps' gradients at lr = 0), shard by shard
    chunk = (N + world - 1) // world
    worst = 0.0
    for nm in NAMES:
        m1 = torch.cat([r["moments"][nm][0] for r in res], 0)
        m2 = torch.cat([r["moments"][nm][1] for r in res], 0)
        assert m1.shape[0] == N and res[1]["moments"][nm][0].shape[0] == min(2 * chunk, N) - chunk
        e1, e2 = _rel(m1, ref["moments"][nm][0]), _rel(m2, ref["moments"][nm][1])
        worst = max(worst, e1, e2)
        assert e1 < 2e-5 and e2 < 4e-5, f"{nm}: moments differ from the single-rank run: {e1:.2e} / {e2:.2e}"
    print(f"[fake world W={world} bsz={bsz} fuse={fuse} live={live} group={group}] worst moment rel err {worst:.2e}; "
          f"{len(fw.log)} collectives", flush=True)


def test_fake_world_detects_collective_order_mismatch(device):
    """the instrument itself: ranks that issue different collectives fail loudly instead of hanging"""
    import torch.distributed as dist
    from fake_world import CollectiveMismatch, FakeWorld

    fw = FakeWorld(2, device)

    def body(rank):
        t = torch.ones(4, device=device)
        if rank == 0:
            dist.all_reduce(t)
        else:
            dist.barrier()
        return t

    with pytest.raises(CollectiveMismatch):
        fw.run(body, timeout=60)


def test_fake_world_collectives_move_the_right_rows(device):
    import torch.distributed as dist
    from fake_world import FakeWorld

    W = 3
    fw = FakeWorld(W, device)

    def body(rank):
        send_splits = [rank + 1 + j for j in range(W)]            # rank r sends r + 1 + j rows to rank j
        recv_splits = [i + 1 + rank for i in range(W)]
        msg = torch.cat([torch.full((n, 2), 100.0 * rank + j, device=device) for j, n in enumerate(send_splits)])
        out = torch.empty((sum(recv_splits), 2), device=device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_to_all_single(out, msg, output_split_sizes=recv_splits, input_split_sizes=send_splits)
        torch.cuda.current_stream().wait_stream(side)
        gathered = torch.empty((W, 2), device=device)
        work = dist.all_gather_into_tensor(gathered, torch.tensor([rank, 10 * rank], device=device,
                                                                  dtype=torch.float32), async_op=True)
        work.wait()
        s = torch.tensor([float(rank + 1)], device=device)
        dist.all_reduce(s)
        return out, recv_splits, gathered, s

    for rank, (out, recv_splits, gathered, s) in enumerate(fw.run(body, timeout=60)):
        want = torch.cat([torch.full((n, 2), 100.0 * i + rank, device=device) for i, n in enumerate(recv_splits)])
        assert torch.equal(out, want)
        assert gathered.tolist() == [[float(i), 10.0 * i] for i in range(W)]
        assert float(s) == 6.0
