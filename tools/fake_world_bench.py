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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bsz", type=int, default=0)
    ap.add_argument("--no-fuse-backward", action="store_true", help="K11 and Adam as two kernels")
    ap.add_argument("--graph", default="off", choices=["off", "on"], help="replay the iteration as one hipGraph")
    ap.add_argument("--profile", action="store_true", help="cProfile the host side of the steps (top functions by own time)")
    a0 = ap.parse_args()

    import bench
    import gaussian_renderer.workload_division as wd
    import utils.general_utils as utils

    install_fake_collectives()
    wd._gather_times_on_host = lambda mine: [list(mine) for _ in range(utils.DEFAULT_GROUP.size())]
    # identical ranks report identical times whatever their band: fed back, that drives the partition to a degenerate
    # one (time / rows rises as a band shrinks).  Keep the even partition: the gather and its bookkeeping still run.
    wd._update_heuristics = lambda *a, **k: None
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    results = []
    for W in a0.worlds:
        rank = W // 2  # a middle band
        # the fields run_workload reads
        a = argparse.Namespace(gaussians=0, width=0, height=0, bsz=a0.bsz, views=8, opacity_logit_mean=0.0,
                               opacity_logit_std=2.0, device_scene=False, no_priming=False,
                               no_fuse_backward=a0.no_fuse_backward, graph=a0.graph, balance_every=0)
        os.environ["WORLD_SIZE"] = str(W)
        utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = (rank if W > 1 else 0), 0, W
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = FakeGroup(W, rank) if W > 1 else utils.SingleGPUGroup()
        wd._BALANCE["mode"] = "pipelined" if W > 1 else "exact"  # (set directly: no gloo group to create here)
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
            # are recorded and the iteration may be captured
            _default_args = utils.default_args
            utils.default_args = lambda **kw: _default_args(**{**kw, "no_heuristics_update": True})
        res = bench.run_workload(a, a0.workload, W, rank if W > 1 else 0, dev, a0.steps, a0.warmup, 1, 0,
                                 single_view=(W == 1), collect_kernels=not a0.profile)
        if a0.profile:
            pr.disable()
            sio = io.StringIO()
            st = pstats.Stats(pr, stream=sio)
            st.sort_stats("tottime").print_stats(30)
            st.sort_stats("cumulative").print_stats("grendel-gs_amd|bench.py|fake_world", 60)
            print(f"# host profile, W={W}: {a0.steps + a0.warmup + 8} steps (incl. priming / warmup)")
            print(sio.getvalue()[:16000], flush=True)
        kern = {k: v["avg_ms"] for k, v in res["kernels"].items()}
        results.append({"world": W, "rank": rank if W > 1 else 0, "ms_per_step": round(res["ms_per_step"], 4),
                        "gaussians_this_rank": res["gaussians_this_rank"], "kernel_ms": kern,
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
