"""The cooperative fake world (tests/fake_world.py) itself, on host tensors: rows land where an all-to-all-v puts them,
process-global state is per rank, mismatched / missing collectives and exceptions fail loudly instead of hanging."""
import os
import sys

import pytest
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_world import CollectiveMismatch, FakeWorld  # noqa: E402


def test_collectives_and_per_rank_globals():
    import utils.general_utils as utils

    W = 3
    fw = FakeWorld(W, "cpu")
    before = utils.GLOBAL_RANK

    def body(rank):
        utils.GLOBAL_RANK = rank
        utils.DEFAULT_GROUP = fw.groups[rank]
        send = [rank + 1 + j for j in range(W)]
        recv = [i + 1 + rank for i in range(W)]
        msg = torch.cat([torch.full((n, 2), 100.0 * rank + j) for j, n in enumerate(send)])
        out = torch.empty((sum(recv), 2))
        dist.all_to_all_single(out, msg, output_split_sizes=recv, input_split_sizes=send)
        assert utils.GLOBAL_RANK == rank and utils.DEFAULT_GROUP.rank() == rank
        g = torch.empty((W, 2))
        dist.all_gather_into_tensor(g, torch.tensor([rank, 10.0 * rank]), async_op=True).wait()
        s = torch.tensor([rank + 1.0])
        dist.all_reduce(s)
        m = torch.tensor([float(rank)])
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        lst = [None] * W
        dist.all_gather_object(lst, rank * 2)
        b = torch.tensor([float(rank)])
        dist.broadcast(b, src=1)
        assert dist.get_world_size() == W and dist.get_rank() == rank and utils.GLOBAL_RANK == rank
        return out, recv, g, s, m, lst, b

    for rank, (out, recv, g, s, m, lst, b) in enumerate(fw.run(body, timeout=30)):
        want = torch.cat([torch.full((n, 2), 100.0 * i + rank) for i, n in enumerate(recv)])
        assert torch.equal(out, want)
        assert g.tolist() == [[float(i), 10.0 * i] for i in range(W)]
        assert float(s) == 6.0 and float(m) == 2.0 and lst == [0, 2, 4] and float(b) == 1.0
    assert utils.GLOBAL_RANK == before  # the process' own state is back
    assert [t for _, t in fw.log][:2] == ["all_to_all_single", "all_to_all_single/read"]


def test_mismatch_deadlock_and_exception_fail_loudly():
    def mismatch(rank):
        t = torch.ones(2)
        dist.all_reduce(t) if rank == 0 else dist.barrier()

    with pytest.raises(CollectiveMismatch, match="calls"):
        FakeWorld(2, "cpu").run(mismatch, timeout=30)

    def missing(rank):
        if rank == 0:
            dist.all_reduce(torch.ones(2))

    with pytest.raises(CollectiveMismatch, match="deadlock"):
        FakeWorld(2, "cpu").run(missing, timeout=30)

    def boom(rank):
        t = torch.ones(2)
        dist.all_reduce(t)
        if rank == 1:
            raise ValueError("boom")
        dist.all_reduce(t)

    with pytest.raises(ValueError, match="boom"):
        FakeWorld(2, "cpu").run(boom, timeout=30)

    def wrong_sizes(rank):
        out = torch.empty((2, 1))
        dist.all_to_all_single(out, torch.ones((3, 1)), output_split_sizes=[1, 1], input_split_sizes=[1, 2])

    with pytest.raises(CollectiveMismatch, match="expects"):
        FakeWorld(2, "cpu").run(wrong_sizes, timeout=30)
