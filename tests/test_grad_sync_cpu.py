"""N2 (replicated-storage gradient sync): all four modes give the dense SUM, world_size 2 and 3, gloo."""
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


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.join(ROOT, "grendel-gs_amd"))
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        sys.path.insert(0, ROOT)
        import diff_gaussian_rasterization as dgr
        from oracle import densify_oracle as DO

        # the HIP row primitives need a GPU: their torch restatements play them in this host-logic test
        dgr.group_rows, dgr.gather_rows = DO.group_rows, DO.gather_rows
        dgr.scatter_rows = lambda order, n, srcs, dsts, row0=0: [d.__setitem__(order[row0:row0 + n].long(), s_[:n]) for s_, d in zip(srcs, dsts)]
        import grad_sync as gs

        dist.init_process_group("gloo", rank=rank, world_size=world)
        N = 500
        shapes = {"_xyz": (N, 3), "_features_dc": (N, 1, 3), "_features_rest": (N, 15, 3), "_opacity": (N, 1),
                  "_scaling": (N, 3), "_rotation": (N, 4)}

        def make(r):
            g = torch.Generator().manual_seed(100 + r)
            vis = torch.rand(N, generator=g) < 0.2  # each rank touches ~20 % of the rows
            m = type("G", (), {})()
            for name, shp in shapes.items():
                p = torch.zeros(shp)
                gr = torch.randn(shp, generator=g)
                gr[~vis] = 0
                p.grad = gr
                setattr(m, name, p)
            return m

        expect = {n: sum(getattr(make(r), n).grad for r in range(world)) for n in shapes}
        for mode in ("dense", "fused_dense", "sparse", "fused_sparse"):
            m = make(rank)
            out = gs.sync_gradients_for_replicated_3dgs_storage(m, dist.group.WORLD, mode, gaussians_distribution=False)
            for n in shapes:
                assert torch.allclose(getattr(m, n).grad, expect[n], atol=1e-6), (mode, n)
            if mode.endswith("sparse"):
                assert 0.2 < out.float().mean().item() < 0.2 * world + 0.05  # union of the ranks' touched rows
        m = make(rank)
        before = m._xyz.grad.clone()
        assert gs.sync_gradients_for_replicated_3dgs_storage(m, dist.group.WORLD, "dense", gaussians_distribution=True) is None
        assert torch.equal(m._xyz.grad, before)  # sharded storage: nothing to synchronise
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_all_sync_modes_equal_dense_sum(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"
