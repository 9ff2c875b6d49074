"""The graphed iteration with the exchange over a one-rank RCCL group, one configuration per process (a crash inside
hipStreamEndCapture must not take a test session with it).  usage: graph_exchange_probe.py <bsz> <overlap 0|1>"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

faulthandler.enable()
bsz, overlap = int(sys.argv[1]), bool(int(sys.argv[2]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29549")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import gaussian_renderer as gr  # noqa: E402
import test_gpu_graphed_step as T  # noqa: E402

dev = torch.device("cuda", 0)
gr.set_exchange_overlap(overlap)
ref = T._train(dev, 8, bsz, graph=False, forced=False)
run = T._train(dev, 8, bsz, graph=True, forced=True)
print("stats", run[3], flush=True)
T._compare(run, ref, 8)
print(f"bsz {bsz} overlap {overlap}: graphed exchange over RCCL equals the eager loop", flush=True)
os._exit(0)  # (destroy_process_group after a captured all-to-all does not return on this stack)
