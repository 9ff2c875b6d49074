"""Host logic added in round 6 that needs no GPU: replay-or-eager of the band-agnostic graph (graphed_step.py) is ONE decision
of the whole group -- two ranks (gloo, world size 2) with DIFFERENT bands of the same partition derive the same graph key and
the same verdict on a captured graph's capacities."""
import os
import socket
import sys
import traceback

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
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import numpy as np

        import gaussian_renderer as gr
        import synthetic_scene as S
        import utils.general_utils as utils
        from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final
        from graphed_step import GraphedIteration, _BandProxy, _Entry

        utils.init_distributed(backend="gloo")
        W, H = 1280, 720
        utils.set_args(utils.default_args(bsz=1))
        utils.set_img_size(H, W)
        cams = S.orbit_cameras(3, W, H)
        hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), world, rank)
        rows = int(utils.TILE_Y)
        ramp = torch.arange(rows, dtype=torch.float32) / rows
        hist.accum_heuristic[cams[1].uid] = 1.0 + 3.0 * ramp          # camera 1: rank 0 gets the taller band
        hist.accum_heuristic[cams[2].uid] = 4.0 - 3.0 * ramp          # camera 2: rank 1 does
        param = torch.nn.Parameter(torch.zeros(8, 3))
        opt = torch.optim.Adam([param], lr=0.0)
        step = GraphedIteration(opt, lambda *a: None)
        out = {}
        parts = []
        for k in range(3):
            strategies, tasks = start_strategy_final([cams[k]], hist)
            parts.append(tuple(strategies[0].division_pos))
            key = step._key([cams[k]], strategies)
            out[f"key{k}"] = repr(key[:3] + key[4:])  # (without the parameter's address, which is the process's own)
            out[f"rows{k}"] = step._band_rows(strategies)
            out[f"mine{k}"] = step._my_bands(strategies)[0]
            # a graph captured for bands of at most 24 rows, no exchange planner capacities yet
            e = _Entry()
            e.sproxies, e.band_cap, e.caps_key = [object()], 24, None
            out[f"usable{k}"] = step._usable(e, [cams[k]], strategies)
        # the exchange planner's capacities against the graph's own layout (identical on every rank: a function of the
        # all-gathered history)
        strategies, _ = start_strategy_final([cams[0]], hist)
        planner = gr._planner(utils.DEFAULT_GROUP, world, 1)
        planner.caps = np.full((world, world, 1), 1024, dtype=np.int64)
        e = _Entry()
        e.sproxies, e.band_cap = [object()], rows
        e.caps_key = np.full((world, world, 1), 1280, dtype=np.int64)
        out["slabs_fit"] = step._usable(e, [cams[0]], strategies)
        planner.caps[1, 0, 0] = 1536
        out["slab_outgrown"] = step._usable(e, [cams[0]], strategies)
        # the stand-in of a strategy during a band-agnostic capture: no host reader may bake a partition in
        px = _BandProxy(strategies[0], 24, None, None, None)
        try:
            px.division_pos
            out["proxy_raises"] = False
        except RuntimeError:
            out["proxy_raises"] = True
        out["proxy_rows"] = px._my_rows()
        out["partitions"] = parts
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


def test_replay_or_eager_is_one_decision_of_the_group():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    a, b = got[0], got[1]
    assert isinstance(a, dict) and isinstance(b, dict), (a, b)
    assert a["partitions"] == b["partitions"] and len(set(a["partitions"])) == 3, a["partitions"]
    for k in range(3):
        assert a[f"mine{k}"] != b[f"mine{k}"]                      # the ranks render different bands ...
        assert a[f"key{k}"] == b[f"key{k}"]                        # ... and still agree on the graph
        assert a[f"rows{k}"] == b[f"rows{k}"] and a[f"usable{k}"] == b[f"usable{k}"]
    assert a["key0"] == a["key1"] == a["key2"]                     # one graph for every partition
    assert a["usable0"] is True and a["usable1"] is False and a["usable2"] is False, a  # (23 / 22 rows; > 24 on ONE rank)
    for g in (a, b):
        assert g["slabs_fit"] is True and g["slab_outgrown"] is False
        assert g["proxy_raises"] is True and g["proxy_rows"] == (-1, 24)


def test_balancer_consumes_a_replays_timestamps_like_event_pairs():
    """workload_division._resolve_deferred_timings: a stats_collector that carries "_gsr_stamps" (what
    GraphedIteration.last_stats holds after a replay) ends up with the same three fields the HIP event pairs of an eager
    iteration fill, and finish_strategy_final turns them into the next per-row costs (workload_division.py:944-998 of the
    reference)"""
    for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gaussian_renderer.workload_division as wd
    import synthetic_scene as S
    import utils.general_utils as utils

    calls = []

    def stamps():
        calls.append(1)
        return {"forward_render_time": 0.30, "backward_render_time": 0.50, "forward_loss_time": 0.10}

    st = {"forward_render_time": 0.0, "backward_render_time": 0.0, "forward_loss_time": 0.0, "_gsr_stamps": stamps}
    wd._resolve_deferred_timings(st)
    assert calls == [1] and "_gsr_stamps" not in st
    assert (st["forward_render_time"], st["backward_render_time"], st["forward_loss_time"]) == (0.30, 0.50, 0.10)

    # rank 0 of two (the peer reports the same times): the resolved times reach the strategy history / the row costs
    class TwoRanks:
        def size(self):
            return 2

        def rank(self):
            return 0

    saved = (utils.GLOBAL_RANK, utils.WORLD_SIZE, utils.DEFAULT_GROUP, utils.our_allgather_among_cpu_processes_float_list,
             wd._BALANCE["mode"])
    try:
        utils.GLOBAL_RANK, utils.WORLD_SIZE = 0, 2
        utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = TwoRanks()
        utils.our_allgather_among_cpu_processes_float_list = lambda data, group: [list(data), list(data)]
        wd._BALANCE["mode"] = "exact"
        utils.set_args(utils.default_args(bsz=1, save_strategy_history=True))
        utils.set_img_size(720, 1280)
        utils.set_cur_iter(10)
        cams = S.orbit_cameras(1, 1280, 720)
        hist = wd.DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 2, 0)
        strategies, _ = wd.start_strategy_final(cams, hist)
        st = {"forward_render_time": 0.0, "backward_render_time": 0.0, "forward_loss_time": 0.0, "_gsr_stamps": stamps}
        wd.finish_strategy_final(cams, hist, strategies, [st])
        assert len(calls) == 2
        rec = hist.history[-1]["batched_camera_info"][0]
        assert abs(rec["each_gpu_running_time"][0] - (0.30 + 0.50 + 2 * 0.10)) < 1e-9, rec
    finally:
        utils.GLOBAL_RANK, utils.WORLD_SIZE, utils.DEFAULT_GROUP = saved[:3]
        utils.IN_NODE_GROUP = saved[2]
        utils.our_allgather_among_cpu_processes_float_list, wd._BALANCE["mode"] = saved[3], saved[4]
