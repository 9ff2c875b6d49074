"""world_size-2 (and 3) `gloo` tests of the multi-GPU path on CPU: partition strategy, the fused sparse
all-to-all exchange (forward ordering + backward scatter-add through the mirror all-to-all), render of
row bands and SUM assembly.  The per-rank device ops (K1, K2, K3-K10) are played by the torch oracle
-- test infrastructure -- because this container has no GPU; everything else is the product's host code
(grendel-gs_amd/gaussian_renderer/*, utils/general_utils.py)."""
import math
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bsz, q):
    try:
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import diff_gaussian_rasterization as dgr
        import synthetic_scene as S
        import utils.general_utils as utils
        from oracle import torch_oracle as O

        utils.init_distributed(backend="gloo")
        utils.set_args(utils.default_args(bsz=bsz, save_strategy_history=(world == 2)))
        # 7 tile rows; 17 for the 4-rank, 33 for the 8-rank partitions (every band needs >= 2 rows)
        N, W, H = 1200, (160 if world < 8 else 96), {2: 112, 3: 112, 4: 272}.get(world, 528)
        utils.set_img_size(H, W)
        utils.set_cur_iter(1)

        # K2 stand-in (the HIP op needs a GPU): same contract, oracle arithmetic
        def k2(image_height, image_width, mp_rank, mp_world_size, means2D, radii, div, cuda_args=None):
            return O.get_local2j_ids_bool(image_height, image_width, mp_world_size, means2D.detach(), radii, div)

        dgr._C.get_local2j_ids_bool = staticmethod(k2)
        from gaussian_renderer import all_to_all_communication_final, get_cuda_args_final
        from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                         set_balance_timing, start_strategy_final)

        set_balance_timing("exact")  # the history is inspected after ONE step below
        cams = S.orbit_cameras(max(bsz, 2), W, H)[:bsz]
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
        if bsz < world:  # skew the cost so that the cut is not in the middle
            hist.accum_heuristic[cams[0].uid][: utils.TILE_Y // 2] = 3.0
        strategies, tasks = start_strategy_final(cams, hist)
        # every tile row of every camera is rendered by exactly one rank
        for k in range(bsz):
            rows = sorted((l, r) for g in range(world) for (kk, l, r) in tasks[g] if kk == k)
            assert rows[0][0] == 0 and rows[-1][1] == utils.TILE_Y
            assert all(a[1] == b[0] for a, b in zip(rows[:-1], rows[1:]))

        full = S.make_gaussians(N, W, H, seed=7, scale_coef=0.015)
        chunk = (N + world - 1) // world
        sl = slice(rank * chunk, min((rank + 1) * chunk, N))
        keys = ["means3D", "scales", "rotations", "shs", "opacities"]
        bg = torch.tensor([0.2, 0.3, 0.4], dtype=torch.float64)
        gen = torch.Generator().manual_seed(3)
        wgts = [torch.rand(3, H, W, generator=gen).double() for _ in range(bsz)]

        def cam_kw(cam):
            return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                        campos=cam.camera_center, W=W, H=H, tanfovx=math.tan(cam.FoVx / 2),
                        tanfovy=math.tan(cam.FoVy / 2), sh_degree=3)

        class R:  # rasterizer duck-type: the exchange only reads raster_settings.image_height/width
            def __init__(self):
                self.raster_settings = type("RS", (), {"image_height": H, "image_width": W})()

        # the device side of the fused exchange (HIP kernels, no GPU here) is played by its torch restatement
        import gaussian_renderer as gr
        from oracle import densify_oracle as DO
        from oracle import exchange_oracle as XO

        dgr.exchange_count, dgr.exchange_pack, dgr.scatter_add_rows = XO.exchange_count, XO.exchange_pack, XO.scatter_add_rows
        dgr.exchange_pack_slab = XO.exchange_pack_slab
        dgr.gather_rows = DO.gather_rows
        dgr.exchange_unpack, dgr.zeros_async = XO.exchange_unpack, XO.zeros_async

        def one_pass(mode):
            """mode: 'reference' = all_to_all_communication_final (the reference-shaped per-camera path),
            'batched' = _batched_exchange_final as ONE exchange with exact sizes, 'pipelined' = one exchange per camera,
            'speculative' = capacity slabs (no read-back of this iteration's counts before the all-to-all; the
            capacities come from the earlier passes), 'overflow' = the same with capacities that are too small: the
            verification must notice and repeat the exchange with exact sizes"""
            mine = {k: v[sl].double().clone().requires_grad_(True) for k, v in full.items()}
            params, cargs = [], []
            for cam, st in zip(cams, strategies):
                m2, rgb, co, radii, depths = O.preprocess(*[mine[k] for k in keys], **cam_kw(cam))
                m2.retain_grad()
                params.append([m2, rgb, co, radii, depths])
                cargs.append(get_cuda_args_final(st, "train"))
            if mode == "reference":
                m2s, rgbs, cos, radiis, depthss, sizes = all_to_all_communication_final([R() for _ in cams], params, cargs,
                                                                                        strategies)
            else:
                gr.set_exchange_overlap(mode == "pipelined")
                gr.set_exchange_grouping(mode != "speculative_per_camera")
                spec = mode in ("speculative", "overflow", "speculative_per_camera")
                renders = [sum(1 for st in strategies if g in st.gpu_ids) for g in range(world)]
                # (set_exchange_grouping: ONE slab exchange for the batch when no rank renders two of its cameras)
                grouped = spec and mode != "speculative_per_camera" and bsz > 1 and max(renders) <= 1
                planner = gr._planner(utils.DEFAULT_GROUP, world, bsz)
                if mode == "overflow":  # every slab one row short of what is needed (where anything is sent)
                    import numpy as np

                    true_sizes = np.asarray(sizes_ref, dtype=np.int64)
                    planner.caps = np.clip(true_sizes - 1, 0, None)
                    planner.caps_list = planner.caps.tolist()
                redone0 = gr.exchange_stats["redone"]

                def run(known=None):
                    return gr._batched_exchange_final(*[[p[c] for p in params] for c in range(5)], [R() for _ in cams],
                                                      strategies, speculate=spec, _known=known)

                m2s, rgbs, cos, radiis, depthss, sizes, (events, token), pending = run()
                assert (pending is not None) == spec
                if spec:  # what render_final does after the renders: look at the counts, repeat if a slab overflowed
                    pl, chunkcnt, counts, lazy, staged = pending
                    m, fitted = pl.resolve(staged)
                    assert m.tolist() == sizes_ref
                    assert fitted == (mode != "overflow")
                    if not fitted:
                        m2s, rgbs, cos, radiis, depthss, sizes, (events, token), pending = run(
                            known=(chunkcnt, counts, m.tolist()))
                        assert pending is None
                    else:
                        sizes = m.tolist()
                        # padding rows: radius 0, behind every source's block; the valid rows are the exact layout's
                        for k in range(bsz):
                            keep = radiis[k] > 0
                            assert int(keep.sum()) == sum(sizes_ref[i][rank][k] for i in range(world))
                            if grouped:  # the whole message (all cameras' slabs) is the local camera's input
                                rows = sum(pl.caps_list[i][rank][kk] for i in range(world) for kk in range(bsz))
                                assert radiis[k].shape[0] == (rows if rank in strategies[k].gpu_ids else 0)
                            else:
                                assert radiis[k].shape[0] == sum(pl.caps_list[i][rank][k] for i in range(world)) or \
                                    rank not in strategies[k].gpu_ids
                assert all(e is None for e in events)  # no side stream on the CPU
                assert (token is not None) == ((mode == "pipelined" or (spec and pending is not None)) and bsz > 1
                                               and not (grouped and pending is not None))
            assert len(sizes) == world and len(sizes[0]) == world and len(sizes[0][0]) == bsz
            loss = torch.zeros((), dtype=torch.float64)
            images = []
            for k, st in enumerate(strategies):
                img = torch.zeros(3, H, W, dtype=torch.float64)
                if rank in st.gpu_ids:
                    mask = st.get_compute_locally()
                    assert int((radiis[k] > 0).sum()) == sum(sizes[i][rank][k] for i in range(world))
                    if m2s[k].shape[0] > 0:
                        img, _, _ = O.render(m2s[k], cos[k], rgbs[k], depthss[k], radiis[k], mask, bg=bg, W=W, H=H)
                    loss = loss + (img * wgts[k]).sum()
                else:
                    assert int((radiis[k] > 0).sum()) == 0
                images.append(img.detach())
            loss = loss + 0.0 * sum(p[0].sum() for p in params)  # keep the graph alive on idle ranks
            if mode != "reference" and token is not None:
                loss = loss + 0.0 * token.sum()  # what the render op of the last local camera does in the product
            loss.backward()
            grads = [mine[k].grad if mine[k].grad is not None else torch.zeros_like(mine[k]) for k in keys]
            m2_grads = [p[0].grad for p in params]  # densification's input survives the exchange
            received = [(m2s[k].detach()[radiis[k] > 0], radiis[k][radiis[k] > 0], depthss[k].detach()[radiis[k] > 0])
                        for k in range(bsz)]
            return images, grads, cargs, m2_grads, received, sizes

        images, grads, cargs, m2g_ref, recv_ref, sizes_ref = one_pass("reference")
        for mode in ("batched", "pipelined", "speculative", "speculative_per_camera", "overflow"):
            im2, gr2, _, m2g, recv2, sizes2 = one_pass(mode)
            assert sizes2 == sizes_ref, mode
            for k in range(bsz):
                assert torch.equal(recv2[k][1], recv_ref[k][1]), f"{mode}: radii / row order of camera {k}"
                assert torch.equal(recv2[k][0], recv_ref[k][0]) and torch.equal(recv2[k][2], recv_ref[k][2]), mode
                assert torch.equal(im2[k], images[k]), mode
                assert (m2g[k] is None) == (m2g_ref[k] is None)
                if m2g[k] is not None:
                    assert torch.allclose(m2g[k], m2g_ref[k], rtol=1e-12, atol=1e-14), mode
            for a, b in zip(gr2, grads):
                assert torch.allclose(a, b, rtol=1e-11, atol=1e-13), f"{mode}: gradients differ from the reference-shaped path"
        stats = [c["stats_collector"] for c in cargs]
        for s_ in stats:
            s_.update(forward_render_time=1.0 + rank, backward_render_time=2.0, forward_loss_time=0.5)
        finish_strategy_final(cams, hist, strategies, stats)
        if world == 2:
            assert len(hist.history) == 1 and len(hist.history[0]["all_gpu_running_time"]) == world
        else:  # timings have no consumer (heuristics frozen, history not saved): no gather, nothing logged
            assert len(hist.history) == 0

        # SUM assembly of the row bands (train_internal.py:466-469)
        stack = torch.stack(images)
        dist.all_reduce(stack)
        # single-process reference on rank 0
        gathered = []
        for gtensor in grads:
            lst = [torch.zeros((min((r + 1) * chunk, N) - r * chunk,) + tuple(gtensor.shape[1:]), dtype=torch.float64)
                   for r in range(world)]
            dist.all_gather(lst, gtensor.contiguous())
            gathered.append(torch.cat(lst, 0))
        if rank == 0:
            ref_in = {k: v.double().clone().requires_grad_(True) for k, v in full.items()}
            ref_loss = 0
            for k, cam in enumerate(cams):
                m2, rgb, co, radii, depths = O.preprocess(*[ref_in[x] for x in keys], **cam_kw(cam))
                mask = torch.ones(utils.TILE_Y, utils.TILE_X, dtype=torch.bool)
                img, _, _ = O.render(m2, co, rgb, depths, radii, mask, bg=bg, W=W, H=H)
                err = (img.detach() - stack[k]).abs().max().item()
                assert err < 1e-9, f"camera {k}: partitioned render differs from the single-rank render by {err}"
                ref_loss = ref_loss + (img * wgts[k]).sum()
            ref_loss.backward()
            for name, gth in zip(keys, gathered):
                ref = ref_in[name].grad
                rel = ((gth - ref).norm() / (ref.norm() + 1e-30)).item()
                assert rel < 1e-9, f"{name}: gradient through the exchange differs, rel {rel}"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,bsz", [(2, 1), (2, 2), (3, 2), (4, 1), (8, 1), (8, 4), (8, 8)])
def test_partitioned_exchange_render_matches_single_rank(world, bsz):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bsz, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_timing_events_only_when_somebody_reads_them():
    """workload_division.timings_have_consumer: the skip rules of the reference's finish_strategy_final
    (workload_division.py:968-978) decide whether the ops record render / loss timing events at all"""
    sys.path.insert(0, os.path.join(ROOT, "grendel-gs_amd"))
    import utils.general_utils as utils
    from gaussian_renderer.workload_division import timings_have_consumer

    class Group:
        def __init__(self, n):
            self.n = n

        def size(self):
            return self.n

    saved = (utils.DEFAULT_GROUP,)
    try:
        utils.set_cur_iter(100)
        utils.set_img_size(1080, 1920)
        utils.set_args(utils.default_args(bsz=1))
        utils.DEFAULT_GROUP = Group(1)
        assert not timings_have_consumer()                     # one rank: nothing to balance
        utils.DEFAULT_GROUP = Group(4)
        assert timings_have_consumer()                         # one 1080p image over four ranks: heuristics live
        utils.set_args(utils.default_args(bsz=4))
        assert not timings_have_consumer()                     # whole images per rank at <= 1080p: frozen
        utils.set_args(utils.default_args(bsz=4, save_strategy_history=True))
        assert timings_have_consumer()                         # ... unless the history is saved
        utils.set_args(utils.default_args(bsz=1))
        utils.set_img_size(500, 800)
        assert not timings_have_consumer()                     # small images: frozen
        utils.set_img_size(2160, 3840)
        utils.set_args(utils.default_args(bsz=1, no_heuristics_update=True))
        assert not timings_have_consumer()
    finally:
        utils.DEFAULT_GROUP = saved[0]
        utils.set_args(utils.default_args(bsz=1))
        utils.set_img_size(1080, 1920)


def test_start_strategy_cut_points():
    """cut-point arithmetic of start_strategy_final against hand-computed cases (workload_division.py:852-941)"""
    sys.path.insert(0, os.path.join(ROOT, "grendel-gs_amd"))
    import synthetic_scene as S
    import utils.general_utils as utils
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, division_pos_heuristic,
                                                     start_strategy_final)

    class G:
        def __init__(self, n):
            self.n = n

        def size(self):
            return self.n

        def rank(self):
            return 0

    utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 4
    utils.DEFAULT_GROUP = G(4)
    utils.set_args(utils.default_args(bsz=1))
    utils.set_img_size(1080, 1920)  # 68 tile rows
    cams = S.orbit_cameras(2, 1920, 1080)
    hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 4, 0)
    st, tasks = start_strategy_final(cams[:1], hist)
    assert st[0].division_pos == [0, 17, 34, 51, 68] and st[0].gpu_ids == [0, 1, 2, 3]
    assert tasks == [[(0, 0, 17)], [(0, 17, 34)], [(0, 34, 51)], [(0, 51, 68)]]
    # bsz 2 on 4 ranks: two ranks per image, cuts snap onto the image border
    utils.set_args(utils.default_args(bsz=2))
    st, tasks = start_strategy_final(cams, hist)
    assert [s.gpu_ids for s in st] == [[0, 1], [2, 3]]
    assert st[0].division_pos == [0, 34, 68] and st[1].division_pos == [0, 34, 68]
    # skewed cost moves the cut; equal-cost rule of division_pos_heuristic
    h = torch.ones(68)
    h[:34] = 3.0
    assert division_pos_heuristic(h, 68, 2, right=True) == [0, 22, 68]  # prefix 3(i+1): first index with prefix > 68 is 22
    utils.DEFAULT_GROUP = utils.SingleGPUGroup()
    utils.WORLD_SIZE = 1


def _gt_worker(rank, world, port, q):
    try:
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import synthetic_scene as S
        import utils.general_utils as utils
        from gaussian_renderer.loss_distribution import (load_camera_from_cpu_to_all_gpu,
                                                         load_camera_from_cpu_to_all_gpu_for_eval)
        from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final

        utils.init_distributed(backend="gloo")
        W, H, bsz = 160, 112, 2
        utils.set_img_size(H, W)
        for storage in (False, True):
            utils.set_args(utils.default_args(bsz=bsz, distributed_dataset_storage=storage))
            cams = S.orbit_cameras(bsz, W, H)
            full = [S.make_gt_image(W, H, seed=40 + k) for k in range(bsz)]
            for k, c in enumerate(cams):
                # distributed storage: only the first rank of the node holds the images (scene/cameras.py:60-75)
                c.original_image_backup = full[k] if (not storage or rank == 0) else None
            hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
            strategies, tasks = start_strategy_final(cams, hist)
            load_camera_from_cpu_to_all_gpu(cams, strategies, tasks)
            for (k, l, r) in tasks[rank]:
                y0, y1 = l * 16, min(r * 16, H)
                assert torch.equal(cams[k].original_image, full[k][:, y0:y1, :]), (storage, k, l, r)
            load_camera_from_cpu_to_all_gpu_for_eval(cams, strategies, tasks)
            for k in range(bsz):
                assert torch.equal(cams[k].original_image, full[k])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


def test_ground_truth_band_staging_local_and_distributed_storage():
    """a15: every rank ends up with exactly the uint8 rows of the bands it renders, with local storage and with
    rank-0-only storage + P2P band transfer (loss_distribution.py:2395-2533); eval gets full images"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("world", [2, 3])
def test_redistribute_gaussians_single_packed_all_to_all(world):
    """N4: every parameter / Adam-moment row reaches its destination rank in ONE all-to-all-v, ordered by
    source rank (scene/gaussian_model.py:1073-1098,1264-1329); row primitives played by their torch restatement"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_workers import redistribution_worker

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=redistribution_worker, args=(r, world, port, False, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def _balance_worker(rank, world, port, q):
    try:
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import synthetic_scene as S
        import utils.general_utils as utils
        from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                         set_balance_timing, start_strategy_final)

        utils.init_distributed(backend="gloo")
        W, H = 1920, 1088  # 68 tile rows; bsz 1 < world and a large image: the heuristics are live
        utils.set_img_size(H, W)
        utils.set_args(utils.default_args(bsz=1, save_strategy_history=True))
        set_balance_timing("pipelined")
        cams = S.orbit_cameras(2, W, H)[:1]
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
        cuts = []
        for it in range(3):
            utils.set_cur_iter(1 + it)
            strategies, tasks = start_strategy_final(cams, hist)
            cuts.append(list(strategies[0].division_pos))
            # rank 0 is three times slower per row than the others
            stats = [{"forward_render_time": 3.0 if rank == 0 else 1.0, "backward_render_time": 0.0, "forward_loss_time": 0.0}]
            finish_strategy_final(cams, hist, strategies, stats)
            assert len(hist.history) == it, "the timings of step t are consumed at step t + 1"
        assert cuts[0] == cuts[1], "step 1 is still cut with the initial heuristics (one-step lag)"
        assert cuts[2][1] < cuts[1][1], "then the slow rank 0 gets a smaller band"
        every = [None] * world
        dist.all_gather_object(every, cuts)
        assert all(c == every[0] for c in every), "every rank computes the same cut points"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


def test_pipelined_load_balancer_lags_one_step_and_needs_no_device_sync():
    """a14 in "pipelined" mode: the previous step's timings, gathered over a host-side gloo group, move the cut points
    one step later; identical on every rank"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_balance_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_local_sampling_strategy_equals_the_reference():
    """a13, `--local_sampling` branch (gaussian_renderer/workload_division.py:858-877 of the reference): every camera
    of the batch is rendered whole by the rank that sampled it.  tests/golden/reference_local_sampling.json holds what
    the REFERENCE's start_strategy_final returned (tests/golden/make_local_sampling_golden.py imports it); the mirror
    must return the same task lists and strategy fields."""
    import json
    from types import SimpleNamespace

    for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import utils.general_utils as utils
    from gaussian_renderer.workload_division import start_strategy_final

    saved = (utils.ARGS, utils.WORLD_SIZE, utils.GLOBAL_RANK, utils.DEFAULT_GROUP, utils.IMG_H, utils.IMG_W,
             utils.TILE_Y, utils.TILE_X)
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_local_sampling.json")))
    assert len(cases) >= 5
    try:
        for c in cases:
            world, bsz, rank = c["world"], c["bsz"], c["rank"]
            utils.set_args(utils.default_args(bsz=bsz, local_sampling=True, distributed_dataset_storage=True))
            utils.WORLD_SIZE, utils.GLOBAL_RANK = world, rank
            utils.DEFAULT_GROUP = SimpleNamespace(size=lambda w=world: w, rank=lambda r=rank: r)
            utils.set_img_size(1080, 1920)
            assert utils.TILE_Y == c["tile_y"]
            cams = [SimpleNamespace(uid=100 + k) for k in range(bsz)]
            strategies, tasks = start_strategy_final(cams, None)
            assert [[list(t) for t in g] for g in tasks] == c["tasks"], c
            got = [{"uid": s.camera.uid, "world_size": s.world_size, "gpu_ids": list(s.gpu_ids),
                    "division_pos": list(s.division_pos), "rank": s.rank} for s in strategies]
            assert got == c["strategies"], (c, got)
    finally:
        (utils.ARGS, utils.WORLD_SIZE, utils.GLOBAL_RANK, utils.DEFAULT_GROUP, utils.IMG_H, utils.IMG_W, utils.TILE_Y,
         utils.TILE_X) = saved


def test_slab_planner_backs_off_after_repeated_redos_and_keeps_pending_per_package():
    """the exchange's capacity planner (host logic): three speculative exchanges in a row that had to be repeated switch
    the next 32 iterations to the exact layout; a package resolves ITS staged counts, whatever was staged after it"""
    import numpy as np

    for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gaussian_renderer as gr

    pl = gr._SlabPlanner(2, 1)
    pl.observe(np.full((2, 2, 1), 100, dtype=np.int64))
    assert pl.backoff == 0
    pl.note(True)
    pl.note(True)
    pl.note(False)          # a success in between resets the streak
    pl.note(True)
    pl.note(True)
    assert pl.backoff == 0
    pl.note(True)
    assert pl.backoff == pl.BACKOFF and pl.redo_streak == 0
    first = pl.stage(torch.full((4, 1), 7, dtype=torch.int32))
    second = pl.stage(torch.full((4, 1), 9, dtype=torch.int32))
    m1, fit1 = pl.resolve(first)       # the older package, resolved after a newer one was staged
    assert int(m1[0, 0, 0]) == 7 and fit1 and pl.pending is second
    m2, _ = pl.resolve(second)
    assert int(m2[0, 0, 0]) == 9 and pl.pending is None


def test_kept_gradient_view_of_a_non_leaf():
    """gaussian_renderer._keep_grad_view: what means2D.retain_grad() is there for (densification reads means2D.grad) minus
    its clone -- the incoming gradient (a column view of K10's [P,9] record) is kept as it is; a tensor used twice
    accumulates like retain_grad(); reading never warns"""
    import warnings

    import torch

    import gaussian_renderer as gr

    class FromRecord(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.sum()

        @staticmethod
        def backward(ctx, g):
            rec = torch.arange(45.0).view(5, 9)
            return rec[:, 0:2]

    p = torch.randn(5, 2, requires_grad=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        t = p * 2.0
        gr._keep_grad_view(t)
        FromRecord.apply(t).backward()
        assert t.grad.stride() == (9, 1) and t.grad._base is not None  # the view, not a copy
        assert torch.equal(t.grad, torch.arange(45.0).view(5, 9)[:, 0:2])
        t2 = p * 3.0
        gr._keep_grad_view(t2)
        (FromRecord.apply(t2) + FromRecord.apply(t2)).backward()
        assert torch.equal(t2.grad, 2 * torch.arange(45.0).view(5, 9)[:, 0:2])
