"""Is an RCCL collective capturable in a hipGraph on this stack?  One-rank group, one collective per process (a crash in
hipStreamEndCapture must not take the other probes with it).  usage: rccl_capture_probe.py {a2a|a2a_uneven|allgather|allreduce}"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
what = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.arange(1024, dtype=torch.float32, device=dev).reshape(256, 4)
y = torch.zeros_like(x)


def op():
    if what == "a2a":
        dist.all_to_all_single(y, x)
    elif what == "a2a_uneven":
        dist.all_to_all_single(y, x, output_split_sizes=[256], input_split_sizes=[256])
    elif what == "allgather":
        dist.all_gather_into_tensor(y, x)
    elif what == "allreduce":
        y.copy_(x)
        dist.all_reduce(y)


op()
torch.cuda.synchronize()
print(what, "eager ok", float(y.sum()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    op()
print(what, "captured", flush=True)
x.mul_(2.0)
g.replay()
torch.cuda.synchronize()
print(what, "replayed", float(y.sum()), "expected", float(x.sum()), flush=True)
dist.destroy_process_group()
