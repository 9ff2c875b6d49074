"""Host-side cost of one training iteration at world size W > 1, measured on ONE GPU: W processes share cuda:0,
talk over gloo (collectives staged through the host by this tool) and run bench.py's step on a tiny scene, so the
step time is ~ the per-rank host overhead of the multi-GPU path (strategy, exchange bookkeeping, autograd, loss).
Usage (GPU box): python tools/multirank_hostprof.py [W] [--profile]"""
import cProfile
import io
import os
import pstats
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, profile, gaussians, size):
    for p in (ROOT, os.path.join(ROOT, "grendel-gs_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dist_workers import stage_collectives_through_host
    import utils.general_utils as utils

    stage_collectives_through_host()
    ar = dist.all_reduce

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, **kw):
        if not t.is_cuda:
            return ar(t, op=op, group=group, **kw)
        c = t.cpu()
        ar(c, op=op, group=group)
        t.copy_(c)

    dist.all_reduce = all_reduce
    orig = utils.init_distributed
    utils.init_distributed = lambda *a, **k: orig(backend="gloo")
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "100", "--warmup", "10", "--no-cpu-baseline",
                "--gaussians", str(gaussians), "--width", str(size), "--height", str(size), "--render-steps", "1",
                "--repeats", "1"] + ([] if os.environ.get("GSR_WITH_EXTRA") else ["--no-extra"]) + (["--workload", os.environ["GSR_WORKLOAD"]] if "GSR_WORKLOAD" in os.environ else []) + os.environ.get("GSR_BENCH_EXTRA", "").split()
    import bench

    if profile and rank == 0:
        pr = cProfile.Profile()
        pr.enable()
        bench.main()
        pr.disable()
        s = io.StringIO()
        st = pstats.Stats(pr, stream=s)
        st.sort_stats("tottime").print_stats(30)
        # what the iteration itself costs: this package's functions by cumulative time, and who calls the tensor methods
        # that may wait for the device
        st.sort_stats("cumtime").print_stats(r"grendel-gs_amd|bench\.py", 45)
        st.print_callers(r"method 'to' of|method 'item' of|method 'cpu' of|method 'tolist' of|synchronize|method 'copy_'")
        print(s.getvalue()[:24000], flush=True)
    else:
        bench.main()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
    profile = "--profile" in sys.argv
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, profile, 4000 * world, 128), nprocs=world, join=True)


if __name__ == "__main__":
    main()
