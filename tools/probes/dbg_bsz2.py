"""debug probe (GPU box): bsz 2 on one rank of a fake 4-rank world, graph vs eager, step by step"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "grendel-gs_amd"), ROOT, os.path.join(ROOT, "tools")]
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import fake_world_bench as fw  # noqa: E402
import test_gpu_dynamic_bands as T  # noqa: E402

fw.install_fake_collectives()
dev = torch.device("cuda:0")
bsz = int(os.environ.get("BSZ", "2"))
import gaussian_renderer as gr  # noqa: E402

gr.set_exchange_grouping(os.environ.get("GROUP", "1") == "1")
T._TRACE = a = []
T._train(dev, fw, 14, graph=False, bsz=bsz)
T._TRACE = b = []
T._train(dev, fw, 14, graph=True, bsz=bsz)
T._TRACE = None
for x, y in zip(a, b):
    print(x[0], x[1], "replayed" if y[2] else "eager   ", "xyz %.9g %.9g  d %.3g   opacity d %.3g" % (x[3], y[3], y[3] - x[3], y[4] - x[4]))
