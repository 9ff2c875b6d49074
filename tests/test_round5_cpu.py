"""Host logic added in round 5 that needs no GPU: the multi-rank decisions of the graphed iteration (gloo, world size 2)
and `bench.py --gpus N` as a supervisor that always prints ONE JSON line."""
import json
import os
import socket
import subprocess
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


def _capture_vote_worker(rank, world, port, q):
    try:
        for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import utils.general_utils as utils
        from graphed_step import GraphedIteration

        utils.init_distributed(backend="gloo")
        votes = [GraphedIteration._group_says_no(False),       # nobody failed
                 GraphedIteration._group_says_no(rank == 1),   # ONE rank's capture failed: everybody must hear it
                 GraphedIteration._group_says_no(True)]
        q.put((rank, votes))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, traceback.format_exc()))


def test_a_failed_capture_disables_replays_on_every_rank():
    """a rank whose capture failed would run eagerly while its peers replay and their collectives would
    no longer pair up -- the decision is one all-reduce over the default group"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_capture_vote_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert got[r] == [False, True, True], got[r]


def test_bench_supervisor_prints_one_json_line_whatever_the_legs_do():
    """`python bench.py --gpus 2` without a launcher supervises its legs as time-limited children: on a box without two
    GPUs both legs die (here: no GPU at all), and the supervisor still prints exactly ONE JSON line that carries the
    metric's fields and an error per leg"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--leg-timeout", "90"], capture_output=True, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["metric"].startswith("training iters/sec") and d["higher_is_better"] is True
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert "error" not in d
    else:
        assert "error" in d and "error" in d["graph_leg"], d
        assert r.returncode != 0
