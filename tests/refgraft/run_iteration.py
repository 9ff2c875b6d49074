"""One process of a level-B1 / B2 graft run: training iterations through the `*_final` call sequence of
train_internal.py:134-208 + the optimizer step of :316-329, on cuda:0, dumping what the iteration produced.

  --side ref     the REFERENCE's own Python (gaussian_renderer, scene.GaussianModel, scene.cameras.Camera,
                 arguments, utils) imported from --ref-root, on THIS repo's operator module + shims
                 (grendel-gs_amd/b1_graft on the path): graft level B1, INTEGRATION.md section 1
  --side mirror  this repo's gaussian_renderer mirror + fused Adam (what bench.py runs)

Both sides execute the SAME function below; only the bootstrap (arguments / model / camera classes) differs.
World size > 1: the GPU box has one device and RCCL refuses two ranks on it, so the ranks share cuda:0, the
process group is gloo and device tensors are staged through the host inside the collectives (patched HERE, in
the test harness -- the product files are untouched).  Test infrastructure; never imported by the product.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class _DoneWork:
    """what an already-completed collective returns for async_op=True"""

    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


def stage_collectives_through_host(dist, torch):
    """device tensors take a round trip through the host inside every collective the path uses"""
    orig = {n: getattr(dist, n) for n in ("all_to_all_single", "all_gather_into_tensor", "all_to_all", "all_reduce",
                                          "scatter", "all_gather", "broadcast")}

    def cpu(t):
        return t.detach().cpu().contiguous()

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, **kw):
        if not output.is_cuda:
            return orig["all_to_all_single"](output, input, output_split_sizes, input_split_sizes, group=group, **kw)
        o = torch.empty(output.shape, dtype=output.dtype)
        orig["all_to_all_single"](o, cpu(input), output_split_sizes, input_split_sizes, group=group)
        output.copy_(o)

    def all_gather_into_tensor(output, input, group=None, **kw):
        if not output.is_cuda:
            return orig["all_gather_into_tensor"](output, input, group=group, **kw)
        o = torch.empty(output.numel(), dtype=output.dtype)  # gloo wants the flat [W * n] form
        orig["all_gather_into_tensor"](o, cpu(input).reshape(-1), group=group)
        output.copy_(o.view(output.shape))
        return _DoneWork() if kw.get("async_op") else None

    def all_to_all(output_tensor_list, input_tensor_list, group=None, **kw):
        # gloo has no list all-to-all: one all_to_all_single over the flattened pieces (only element counts matter, as
        # with RCCL: the reference sends [n,1,9] pieces into [n,9] buffers, gaussian_renderer/__init__.py:592-603)
        ins = [cpu(t).reshape(-1) for t in input_tensor_list]
        dtype = output_tensor_list[0].dtype
        o = torch.empty(sum(t.numel() for t in output_tensor_list), dtype=dtype)
        orig["all_to_all_single"](o, torch.cat(ins) if ins else o[:0], [t.numel() for t in output_tensor_list],
                                  [t.numel() for t in ins], group=group)
        off = 0
        for d in output_tensor_list:
            d.copy_(o[off:off + d.numel()].view(d.shape))
            off += d.numel()

    def all_reduce(tensor, *a, **kw):
        if not tensor.is_cuda:
            return orig["all_reduce"](tensor, *a, **kw)
        t = cpu(tensor)
        orig["all_reduce"](t, *a, **kw)
        tensor.copy_(t)

    def scatter(tensor, scatter_list=None, src=0, group=None, **kw):
        if not tensor.is_cuda:
            return orig["scatter"](tensor, scatter_list, src, group=group, **kw)
        t = torch.empty(tensor.shape, dtype=tensor.dtype)
        orig["scatter"](t, None if scatter_list is None else [cpu(x) for x in scatter_list], src, group=group)
        tensor.copy_(t)

    def all_gather(tensor_list, tensor, group=None, **kw):
        if not tensor.is_cuda:
            return orig["all_gather"](tensor_list, tensor, group=group, **kw)
        outs = [torch.empty(t.shape, dtype=t.dtype) for t in tensor_list]
        orig["all_gather"](outs, cpu(tensor), group=group)
        for d, s in zip(tensor_list, outs):
            d.copy_(s)

    def broadcast(tensor, src=0, group=None, **kw):
        if not tensor.is_cuda:
            return orig["broadcast"](tensor, src, group=group, **kw)
        t = cpu(tensor)
        orig["broadcast"](t, src, group=group)
        tensor.copy_(t)

    for n, f in dict(all_to_all_single=all_to_all_single, all_gather_into_tensor=all_gather_into_tensor,
                     all_to_all=all_to_all, all_reduce=all_reduce, scatter=scatter, all_gather=all_gather,
                     broadcast=broadcast).items():
        setattr(dist, n, f)
        setattr(torch.distributed, n, f)
    # torch.distributed.nn.functional._AlltoAll emulates the all-to-all with equal-size scatters on gloo; report "nccl"
    # so that it takes the dist.all_to_all branch (staged above), as it does on the real RCCL group
    dist.get_backend = lambda group=None: "nccl"
    torch.distributed.get_backend = dist.get_backend


def bootstrap_reference(a, scene):
    """the reference's own start-up (train.py:33-79), minus the dataset: argument groups, init_distributed, init_args,
    block / image size, timers, log file"""
    import torch
    import torch.distributed as dist

    if int(scene["world"]) > 1:
        real_init = dist.init_process_group

        def gloo_init(backend=None, *args, **kw):  # utils/general_utils.py:200 hard-codes "nccl"
            return real_init("gloo", *args, **kw)

        dist.init_process_group = gloo_init
        torch.distributed.init_process_group = gloo_init
        stage_collectives_through_host(dist, torch)
    from argparse import ArgumentParser

    import diff_gaussian_rasterization
    import utils.general_utils as utils
    from arguments import (AuxiliaryParams, BenchmarkParams, DebugParams, DistributionParams, ModelParams,
                           OptimizationParams, PipelineParams, init_args)
    from utils.timer import Timer

    assert utils.__file__.startswith(a.ref_root), utils.__file__
    parser = ArgumentParser()
    AuxiliaryParams(parser)
    lp, op, pp = ModelParams(parser), OptimizationParams(parser), PipelineParams(parser)
    DistributionParams(parser)
    BenchmarkParams(parser)
    DebugParams(parser)
    argv = ["-s", "/tmp/none", "--model_path", a.workdir, "--bsz", str(int(scene["bsz"])), "--preload_dataset_to_gpu"]
    if int(scene["fake_times"]):
        argv += ["--save_strategy_history"]
    args = parser.parse_args(argv)
    torch.cuda.set_device(0)
    utils.init_distributed(args)
    init_args(args)
    args = utils.get_args()
    utils.set_block_size(*diff_gaussian_rasterization._C.get_block_XY())  # arguments/__init__.py:254-257
    utils.set_img_size(int(scene["height"]), int(scene["width"]))
    utils.set_log_file(open(os.path.join(a.workdir, f"python_rk={utils.GLOBAL_RANK}.log"), "w"))
    utils.set_timers(Timer(args))
    return utils, args, op.extract(args), pp.extract(args)


def bootstrap_mirror(a, scene):
    import torch

    import utils.general_utils as utils

    assert utils.__file__.startswith(os.path.join(ROOT, "grendel-gs_amd")), utils.__file__
    torch.cuda.set_device(0)
    world = int(scene["world"])
    if world > 1:
        import torch.distributed as dist

        utils.init_distributed(backend="gloo")
        stage_collectives_through_host(dist, torch)
    else:
        utils.init_distributed()
    args = utils.default_args(bsz=int(scene["bsz"]), save_strategy_history=bool(int(scene["fake_times"])),
                              log_folder=a.workdir, gaussians_distribution=world > 1, image_distribution=world > 1)
    utils.set_args(args)
    utils.set_img_size(int(scene["height"]), int(scene["width"]))
    pipe = type("Pipe", (), {"debug": False})()
    return utils, args, None, pipe


def build_model(side, scene, utils, opt_args):
    import torch

    n = scene["xyz"].shape[0]
    W, r = utils.DEFAULT_GROUP.size(), utils.DEFAULT_GROUP.rank()
    chunk = (n + W - 1) // W  # utils/general_utils.py:272-276 get_local_chunk_l_r, the shard a rank owns at start-up
    lo, hi = r * chunk, min((r + 1) * chunk, n)
    names = {"_xyz": "xyz", "_features_dc": "features_dc", "_features_rest": "features_rest", "_scaling": "scaling",
             "_rotation": "rotation", "_opacity": "opacity"}
    if side == "ref":
        from scene.gaussian_model import GaussianModel

        pc = GaussianModel(3)
        pc.spatial_lr_scale = 1.0
    else:
        import synthetic_scene as S

        pc = S.SyntheticGaussianModel.__new__(S.SyntheticGaussianModel)
        torch.nn.Module.__init__(pc)
        pc.max_sh_degree = 3
    pc.active_sh_degree = 3
    for attr, key in names.items():
        t = torch.from_numpy(np.ascontiguousarray(scene[key][lo:hi])).cuda()
        setattr(pc, attr, torch.nn.Parameter(t.requires_grad_(True)))
    if side == "ref":
        pc.training_setup(opt_args)  # scene/gaussian_model.py:244-333: the reference's Adam groups and schedules
    else:
        import math

        from fused_optim import FusedAdam

        bsz = utils.get_args().bsz
        groups = pc.param_groups()
        opt = FusedAdam(groups, lr=0.0, eps=1e-15)
        for g in opt.param_groups:  # lr_scale_mode "sqrt" (scene/gaussian_model.py:300-312)
            s = math.sqrt(bsz)
            g["lr"] *= s
            g["eps"] /= s
            g["betas"] = [b ** bsz for b in g["betas"]]
        pc.optimizer = opt
    return pc, (lo, hi)


def build_cameras(side, scene):
    import torch

    cams = []
    for k in range(scene["cam_R"].shape[0]):
        img = torch.from_numpy(np.ascontiguousarray(scene["gt"][k]))
        R, T = scene["cam_R"][k].astype(np.float64), scene["cam_T"][k].astype(np.float64)
        if side == "ref":
            from scene.cameras import Camera

            cams.append(Camera(colmap_id=k, R=R, T=T, FoVx=float(scene["cam_fovx"][k]), FoVy=float(scene["cam_fovy"][k]),
                               image=img, gt_alpha_mask=None, image_name=f"v{k:03d}", uid=k))
        else:
            import synthetic_scene as S

            W, H = int(scene["width"]), int(scene["height"])
            cam = S.SyntheticCamera(k, W, H, R=torch.from_numpy(R), T=torch.from_numpy(T), device="cuda")
            cam.FoVx, cam.FoVy = float(scene["cam_fovx"][k]), float(scene["cam_fovy"][k])
            cam.original_image_backup = img.cuda()
            cams.append(cam)
    return cams


def fake_time(rank, cam_uid, rows, iteration):
    """deterministic stand-in for the measured (render fwd, render bwd, loss fwd) ms of a band"""
    base = 0.01 * rows * (1.0 + 0.6 * rank) * (1.0 + 0.1 * ((iteration + cam_uid) % 3))
    return {"forward_render_time": base, "backward_render_time": 2.0 * base, "forward_loss_time": 0.25 * base}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["ref", "mirror"], required=True)
    ap.add_argument("--scene", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--ref-root", default=os.path.join(ROOT, "_refstage", "reference"))
    ap.add_argument("--workdir", default="/tmp/refgraft")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    if a.side == "ref":
        paths = [a.ref_root, os.path.join(ROOT, "grendel-gs_amd", "b1_graft")]
    else:
        paths = [os.path.join(ROOT, "grendel-gs_amd")]
    for p in reversed(paths):
        sys.path.insert(0, p)
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") not in (HERE, ROOT)]  # no accidental `utils` shadowing

    import torch
    import torch.distributed as dist

    scene = dict(np.load(a.scene))
    utils, args, opt_args, pipe = (bootstrap_reference if a.side == "ref" else bootstrap_mirror)(a, scene)

    import gaussian_renderer
    from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final
    from gaussian_renderer.loss_distribution import batched_loss_computation, load_camera_from_cpu_to_all_gpu
    from gaussian_renderer.workload_division import (DivisionStrategyHistoryFinal, finish_strategy_final,
                                                     start_strategy_final)
    want = a.ref_root if a.side == "ref" else os.path.join(ROOT, "grendel-gs_amd")
    assert gaussian_renderer.__file__.startswith(want), gaussian_renderer.__file__
    import diff_gaussian_rasterization as dgr

    assert dgr.__file__.startswith(os.path.join(ROOT, "grendel-gs_amd")), dgr.__file__
    if a.side == "mirror":
        from gaussian_renderer.workload_division import set_balance_timing

        set_balance_timing("exact")

    world, rank = utils.DEFAULT_GROUP.size(), utils.DEFAULT_GROUP.rank()
    bsz, iters = int(scene["bsz"]), int(scene["iters"])
    H, W_img = int(scene["height"]), int(scene["width"])
    pc, (lo, hi) = build_model(a.side, scene, utils, opt_args)
    cams = build_cameras(a.side, scene)
    dataset = type("D", (), {"cameras": cams})()
    hist = DivisionStrategyHistoryFinal(dataset, world, rank)
    if "heuristic" in scene:
        for k, c in enumerate(cams):
            cur = hist.accum_heuristic[c.uid]
            hist.accum_heuristic[c.uid] = torch.from_numpy(scene["heuristic"][k]).to(cur.device)
    bg = torch.from_numpy(scene["bg"]).cuda()

    out = {}
    iteration = 1
    for it in range(iters):
        utils.set_cur_iter(iteration)
        if a.side == "ref":
            pc.update_learning_rate(iteration)  # train_internal.py:103
        else:  # the same schedule (utils/general_utils.py:364-397 with scene/gaussian_model.py:320-331's arguments)
            s = float(np.sqrt(bsz))
            t = np.clip(iteration / 30000.0, 0, 1)
            pc.optimizer.param_groups[0]["lr"] = float(np.exp(np.log(0.00016 * s) * (1 - t) + np.log(0.0000016 * s) * t))
        batch = [cams[(it * bsz + j) % len(cams)] for j in range(bsz)]
        with torch.no_grad():
            strategies, tasks = start_strategy_final(batch, hist)
            load_camera_from_cpu_to_all_gpu(batch, strategies, tasks)
        pkg = distributed_preprocess3dgs_and_all2all_final(batch, pc, pipe, bg, batched_strategies=strategies,
                                                           mode="train")
        images, masks = render_final(pkg, strategies)
        stats = [ca["stats_collector"] for ca in pkg["batched_cuda_args"]]
        loss, parts = batched_loss_computation(images, batch, masks, strategies, stats)
        loss.backward()
        if int(scene["fake_times"]):
            for st, strat, cam in zip(stats, strategies, batch):
                for k in ("_fwd_events", "_bwd_events", "_loss_events"):
                    st.pop(k, None)
                if rank in strat.gpu_ids:
                    j = strat.gpu_ids.index(rank)
                    st.update(fake_time(rank, cam.uid, strat.division_pos[j + 1] - strat.division_pos[j], it))
        with torch.no_grad():
            finish_strategy_final(batch, hist, strategies, stats)
        torch.cuda.synchronize()

        tag = f"it{it}_"
        out[tag + "cuts"] = np.array([x for s in strategies for x in ([-1] + list(s.gpu_ids) + [-2] + list(s.division_pos))],
                                     np.int64)
        out[tag + "tasks"] = np.array([x for g in tasks for t in g for x in t] or [0], np.int64)
        out[tag + "sizes"] = np.array([[list(b) for b in a_] for a_ in pkg["gpui_to_gpuj_imgk_size"]], np.int64)
        out[tag + "loss"] = np.float64(loss.item())
        out[tag + "parts"] = np.array([[float(p[0]), float(p[1])] for p in parts], np.float64)
        last = it == iters - 1
        if last:
            stack = torch.zeros(bsz, 3, H, W_img)
            for k, img in enumerate(images):
                if img is not None and img.dim() == 3:
                    stack[k] = img.detach().cpu()
            tot = torch.tensor([loss.item()], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(stack)  # train_internal.py:466-469: images are assembled by SUM
                dist.all_reduce(tot)
            out["images"] = stack.numpy()
            out["loss_total"] = np.float64(tot.item())
            for k in range(bsz):
                m2 = pkg["batched_locally_preprocessed_mean2D"][k]
                out[f"means2D_grad_{k}"] = (torch.zeros_like(m2) if m2.grad is None else m2.grad).detach().cpu().numpy()
                out[f"radii_{k}"] = pkg["batched_locally_preprocessed_radii"][k].detach().cpu().numpy()
                if images[k] is not None:  # the rows this rank rendered camera k from, in arrival order (row a6);
                    # the mirror's capacity slabs carry padding rows (radius 0) between the sources' blocks
                    keep = pkg["batched_radii_redistributed"][k] > 0
                    out[f"recv_means2D_{k}"] = pkg["batched_means2D_redistributed"][k].detach()[keep].cpu().numpy()
                    out[f"recv_depths_{k}"] = pkg["batched_depths_redistributed"][k].detach()[keep].cpu().numpy()
            for attr in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
                g = getattr(pc, attr).grad
                out["grad" + attr] = (torch.zeros_like(getattr(pc, attr)) if g is None else g).detach().cpu().numpy()
        # optimizer step (train_internal.py:316-329)
        with torch.no_grad():
            if a.side == "ref":
                for p in pc.all_parameters():
                    if p.grad is not None:
                        p.grad /= args.bsz
                pc.optimizer.step()
                pc.optimizer.zero_grad(set_to_none=True)
            else:
                pc.optimizer.step(grad_scale=1.0 / args.bsz)
                pc.optimizer.zero_grad(set_to_none=True)
        iteration += bsz
    out["xyz_lr"] = np.float64(pc.optimizer.param_groups[0]["lr"])
    for attr in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        out["param" + attr] = getattr(pc, attr).detach().cpu().numpy()
    out["heuristic_final"] = np.stack([hist.accum_heuristic[c.uid].detach().cpu().numpy() for c in cams])
    out["history_len"] = np.int64(len(hist.history))
    out["shard"] = np.array([lo, hi], np.int64)
    np.savez(a.out, **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("ok", flush=True)


if __name__ == "__main__":
    main()
