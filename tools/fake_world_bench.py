"""What ONE rank of a W-GPU run does per iteration, measured on ONE MI355X without the other W - 1 GPUs.

The build box has a single GPU and RCCL refuses two ranks on one device, so no multi-GPU number can be measured here.
This tool runs bench.py's training step as rank r of a FAKE world of W identical ranks inside one process: the process
group is a stand-in whose collectives are device-local and asynchronous --
  all_gather_into_tensor : every rank reports the counts this rank has (identical shards);
  all_to_all_single      : what this rank receives from source i is what it sends to itself (identical shards), so the
                           received tensors have the size and the statistics a real rank's have; the mirror all-to-all of
                           the backward likewise;
  barrier / all_reduce   : nothing
-- so the step contains everything a real rank executes (K1 on N / W Gaussians, count, pack into capacity slabs, unpack,
K3-K10 on its row band of the Gaussians of ALL ranks, band loss, scatter-add, K11, Adam on its shard, the host-side
partition / verification / autograd bookkeeping) EXCEPT the wire time of the two all-to-alls and load imbalance.  It
answers the two questions the missing hardware leaves open: is the W > 1 step host-bound, and what is the upper bound
of the pixel-partition speed-up (t(W = 1, whole scene) / t(one rank of W)).

Usage (GPU box): python tools/fake_world_bench.py [--workload c2] [--worlds 1 2 4 8] [--steps 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # (see bench.py: must precede the import of torch)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class FakeGroup:
    def __init__(self, world, rank):
        self.world, self.r = world, rank

    def size(self):
        return self.world

    def rank(self):
        return self.r


class _Done:
    def wait(self, *a, **k):
        return True


def install_fake_collectives():
    def all_gather_into_tensor(output, input, group=None, async_op=False, **kw):
        W = group.size()
        output.view(W, -1).copy_(input.reshape(1, -1).expand(W, -1))
        return _Done() if async_op else None

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, **kw):
        W, me = group.size(), group.rank()
        lo = sum(input_split_sizes[:me])
        seg = input[lo:lo + input_split_sizes[me]]          # what this rank sends to itself
        n = seg.shape[0]
        if n > 0 and all(k == n for k in output_split_sizes):
            # the steady state (identical ranks report identical counts, so every source's segment has this size): ONE
            # broadcast copy -- a real all-to-all is one RCCL kernel, not W copies (round 4: the W-copy form made the
            # instrument's own launches 16 of the 58 nodes of a replayed iteration)
            output.view(W, n, -1).copy_(seg.reshape(1, n, -1).expand(W, n, -1))
            return
        o = 0
        for i in range(W):                                  # every source looks like this rank
            n = output_split_sizes[i]
            m = min(n, seg.shape[0])
            output[o:o + m].copy_(seg[:m])
            if n > m:
                output[o + m:o + n].zero_()
            o += n

    dist.all_gather_into_tensor = all_gather_into_tensor
    dist.all_to_all_single = all_to_all_single
    dist.barrier = lambda *a, **k: None
    dist.all_reduce = lambda *a, **k: None


# ---- a BALANCED partition for the fake world (round 4).  With the even partition the instrument's middle rank renders
# the densest band of the image: 1 / 5.3 of the composite work instead of 1 / 8 on configs[2]'s shape
# (tools/diag/wg_timeline.py) -- a real run's load balancer shrinks that band.  Grendel's balancer is a fixed-point
# iteration on per-row costs (new cost of a band's rows = the band's measured time / its rows, workload_division.py:
# 953-998); it is run here round by round: in a round every rank position r = 0 .. W-1 is measured in turn on the
# CURRENT partition (heuristics frozen inside the round), then the per-row costs are updated from all ranks' times with
# the reference's formula.  The final measurement (optionally as a hipGraph) uses the converged partition, frozen.
_SHARED = {"heur": {}, "round": {}}


def _install_balancer_hooks(wd, utils):
    import torch as _torch

    init0 = wd.DivisionStrategyHistoryFinal.__init__

    def init(self, dataset, world_size, rank):
        init0(self, dataset, world_size, rank)
        for cam in dataset.cameras:
            if cam.uid in _SHARED["heur"]:
                self.accum_heuristic[cam.uid] = _SHARED["heur"][cam.uid].clone()

    wd.DivisionStrategyHistoryFinal.__init__ = init
    my0 = wd._my_times

    def my_times(batched_strategies, batched_statistic_collector):
        mine = my0(batched_strategies, batched_statistic_collector)
        for k, st in enumerate(batched_strategies):
            if mine[k] >= 0:
                rec = _SHARED["round"].setdefault(st.camera.uid, {})
                rec.setdefault(utils.GLOBAL_RANK, []).append((mine[k], tuple(st.gpu_ids), tuple(st.division_pos)))
        return mine

    wd._my_times = my_times
    _SHARED["torch"] = _torch


def _end_of_round(tile_y, per_camera=False):
    """per-row costs of every camera from the round's measurements (all rank positions), the reference's update rule"""
    torch_ = _SHARED["torch"]
    spread = []
    for uid, per_rank in _SHARED["round"].items():
        any_rec = next(iter(per_rank.values()))[-1]
        gpu_ids, div = any_rec[1], any_rec[2]
        new = torch_.ones((tile_y,), dtype=torch_.float32)
        times = []
        for j, g in enumerate(gpu_ids):
            recs = [t for (t, gi, dv) in per_rank.get(g, []) if dv == div]
            if not recs:
                continue
            t = sorted(recs)[len(recs) // 2]  # median over the round's visits of this camera
            times.append(t)
            new[div[j]:div[j + 1]] = t / (div[j + 1] - div[j])
        _SHARED["heur"][uid] = new
        if times:
            spread.append(max(times) / (sum(times) / len(times)))
    # ONE partition for all cameras (default): the mean of the per-camera row costs.  Rounds 3-5 a captured iteration
    # baked its partition in (grid sizes, band rows of the loss), so per-camera cut points needed one graph per camera;
    # --per-camera keeps every camera's own costs, as the reference does (workload_division.py:806-849) -- since round 6
    # the bands are device data and one graph serves them all (graphed_step.py)
    if not per_camera:
        mean = sum(_SHARED["heur"].values()) / max(len(_SHARED["heur"]), 1)
        for uid in list(_SHARED["heur"]):
            _SHARED["heur"][uid] = mean.clone()
    _SHARED["round"] = {}
    return sum(spread) / max(len(spread), 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bsz", type=int, default=0)
    ap.add_argument("--no-fuse-backward", action="store_true", help="K11 and Adam as two kernels")
    ap.add_argument("--graph", default="off", choices=["off", "on"], help="replay the iteration as one hipGraph")
    ap.add_argument("--balanced", type=int, default=0, metavar="ROUNDS",
                    help="balance the row partition with Grendel's own rule over ROUNDS rounds (every rank position "
                         "measured in turn), then measure EVERY rank position on the converged partition")
    ap.add_argument("--profile", action="store_true", help="cProfile the host side of the steps (top functions by own time)")
    ap.add_argument("--live-timings", action="store_true",
                    help="with --graph on: the heuristics stay live, i.e. the replays carry the device timestamps that feed "
                         "finish_strategy_final (the update itself stays switched off in this instrument, see below)")
    ap.add_argument("--per-camera", action="store_true",
                    help="with --balanced: every camera keeps its OWN converged cut points (the reference's behaviour)")
    a0 = ap.parse_args()

    import bench
    import gaussian_renderer.workload_division as wd
    import utils.general_utils as utils

    install_fake_collectives()
    wd._gather_times_on_host = lambda mine: [list(mine) for _ in range(utils.DEFAULT_GROUP.size())]
    utils.our_allgather_among_cpu_processes_float_list = lambda data, group: [list(data) for _ in range(group.size())]
    # identical ranks report identical times whatever their band: fed back, that drives the partition to a degenerate
    # one (time / rows rises as a band shrinks).  The product's per-iteration update is therefore switched off; the
    # even partition stays unless --balanced runs the same rule round by round over all rank positions (above).
    wd._update_heuristics = lambda *a, **k: None
    if a0.balanced:
        _install_balancer_hooks(wd, utils)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    results = []
    def one_run(W, rank, graph, steps, warmup, frozen, timed_probe=False, collect=True):
        a = argparse.Namespace(gaussians=0, width=0, height=0, bsz=a0.bsz, views=8, opacity_logit_mean=0.0,
                               opacity_logit_std=2.0, device_scene=False, no_priming=False,
                               no_fuse_backward=a0.no_fuse_backward, graph=graph, balance_every=0)
        os.environ["WORLD_SIZE"] = str(W)
        utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = (rank if W > 1 else 0), 0, W
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = FakeGroup(W, rank) if W > 1 else utils.SingleGPUGroup()
        wd._BALANCE["mode"] = "exact"
        import gaussian_renderer as gr

        gr._PLANNERS.clear()
        _default_args = utils.default_args
        # frozen: nothing consumes timings (graph-capable); timed_probe: the ops record their render / loss events
        utils.default_args = lambda **kw: _default_args(**{**kw, "no_heuristics_update": frozen,
                                                           "save_strategy_history": timed_probe})
        try:
            return bench.run_workload(a, a0.workload, W, rank if W > 1 else 0, dev, steps, warmup, 1, 0,
                                      single_view=(W == 1), collect_kernels=collect)
        finally:
            utils.default_args = _default_args

    def bench_bands(res):
        ex = res.get("exchange") or {}
        return ex.get("bands_last_step")

    if a0.balanced:
        base = one_run(1, 0, a0.graph, a0.steps, a0.warmup, True)
        print(json.dumps({"world": 1, "ms_per_step": round(base["ms_per_step"], 4)}), flush=True)
        for W in [w for w in a0.worlds if w > 1]:
            _SHARED["heur"].clear()
            for rnd in range(a0.balanced):
                for r in range(W):
                    one_run(W, r, "off", 8, 2, True, timed_probe=True, collect=False)
                imb = _end_of_round(utils.TILE_Y, a0.per_camera)
                print(f"# W={W} round {rnd}: max / mean band time of the measured partition {imb:.3f}", flush=True)
            per_rank = []
            for r in range(W):
                res = one_run(W, r, a0.graph, a0.steps, a0.warmup, True)
                per_rank.append({"rank": r, "ms_per_step": round(res["ms_per_step"], 4), "graph": res.get("graph"),
                                 "bands": bench_bands(res),
                                 "kernel_sum_ms": round(sum(v["avg_ms"] for v in res["kernels"].values()), 4),
                                 "render_ms": round(sum(v["avg_ms"] for k, v in res["kernels"].items()
                                                        if k.startswith(("binning", "composite", "l1_ssim"))), 4)})
            worst = max(p_["ms_per_step"] for p_ in per_rank)
            print(json.dumps({"world": W, "balanced_rounds": a0.balanced, "graph": a0.graph, "per_rank": per_rank,
                              "step_ms_max_over_ranks": worst,
                              "speedup_bound": round(base["ms_per_step"] / worst, 3)}), flush=True)
        return

    for W in a0.worlds:
        rank = W // 2  # a middle band
        # the fields run_workload reads
        a = argparse.Namespace(gaussians=0, width=0, height=0, bsz=a0.bsz, views=8, opacity_logit_mean=0.0,
                               opacity_logit_std=2.0, device_scene=False, no_priming=False,
                               no_fuse_backward=a0.no_fuse_backward, graph=a0.graph, balance_every=0)
        os.environ["WORLD_SIZE"] = str(W)
        utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = (rank if W > 1 else 0), 0, W
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = FakeGroup(W, rank) if W > 1 else utils.SingleGPUGroup()
        # (set directly: no gloo group to create here; GSR_FAKE_BALANCE_MODE=exact: the reference's schedule -- the host
        # waits for an iteration's times before it starts the next)
        wd._BALANCE["mode"] = os.environ.get("GSR_FAKE_BALANCE_MODE", "pipelined") if W > 1 else "exact"
        import gaussian_renderer as gr

        gr._PLANNERS.clear()
        for k in gr.exchange_stats:
            gr.exchange_stats[k] = 0
        if a0.profile:
            import cProfile
            import io
            import pstats

            pr = cProfile.Profile()
            pr.enable()
        if a0.graph == "on":
            # the instrument keeps the even partition anyway (above); say so to the mirror, so that no timing events
            # are recorded (--live-timings: leave the heuristics live -- the replays then carry device timestamps)
            _default_args = utils.default_args
            utils.default_args = lambda **kw: _default_args(**{**kw, "no_heuristics_update": not a0.live_timings})
        res = bench.run_workload(a, a0.workload, W, rank if W > 1 else 0, dev, a0.steps, a0.warmup, 1, 0,
                                 single_view=(W == 1), collect_kernels=not a0.profile)
        if a0.profile:
            pr.disable()
            sio = io.StringIO()
            st = pstats.Stats(pr, stream=sio)
            st.sort_stats("tottime").print_stats(45)
            st.sort_stats("cumulative").print_stats("grendel-gs_amd|bench.py|fake_world", 70)
            print(f"# host profile, W={W}: {a0.steps + a0.warmup + 8} steps (incl. priming / warmup)")
            print(sio.getvalue()[:24000], flush=True)
        kern = {k: v["avg_ms"] for k, v in res["kernels"].items()}
        import diff_gaussian_rasterization as _dgr_

        results.append({"world": W, "rank": rank if W > 1 else 0, "ms_per_step": round(res["ms_per_step"], 4),
                        "gaussians_this_rank": res["gaussians_this_rank"],
                        "pairs_last_view": int(getattr(_dgr_._RenderGaussians, "last_num_rendered", 0) or 0),
                        "kernel_ms": kern,
                        "kernel_sum_ms": round(sum(kern.values()), 4), "exchange_layouts": dict(gr.exchange_stats),
                        "graph": res.get("graph")})
        if a0.graph == "on":
            utils.default_args = _default_args
        print(json.dumps(results[-1]), flush=True)
    base = next((r for r in results if r["world"] == 1), None)
    if base:
        print("# upper bound of the pixel-partition speed-up (no wire time, no imbalance): " + ", ".join(
            f"W={r['world']}: {base['ms_per_step'] / r['ms_per_step']:.2f}x" for r in results if r["world"] > 1))


if __name__ == "__main__":
    main()
