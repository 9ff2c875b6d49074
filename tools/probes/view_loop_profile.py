"""host profile of the forward-only view loop (GPU box): python tools/probes/view_loop_profile.py [views]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "grendel-gs_amd"), ROOT]
import torch  # noqa: E402

import synthetic_scene as S  # noqa: E402
import utils.general_utils as utils  # noqa: E402
from gaussian_renderer import distributed_preprocess3dgs_and_all2all_final, render_final, settle  # noqa: E402
from gaussian_renderer.workload_division import DivisionStrategyHistoryFinal, start_strategy_final  # noqa: E402

dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1920, 1080
utils.GLOBAL_RANK, utils.LOCAL_RANK, utils.WORLD_SIZE = 0, 0, 1
utils.DEFAULT_GROUP = utils.IN_NODE_GROUP = utils.SingleGPUGroup()
utils.set_args(utils.default_args(bsz=1))
utils.set_img_size(H, W)
utils.set_cur_iter(1)
model = S.SyntheticGaussianModel(N, W, H, seed=0, device=dev, on_device=True)
cams = S.orbit_cameras(8, W, H, device=dev)
hist = DivisionStrategyHistoryFinal(S.SyntheticDataset(cams), 1, 0)
bg = torch.zeros(3, device=dev)
pipe = type("P", (), {"debug": False})()
side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
pending = [[], []]
turn = [0]


def view():
    k = turn[0] & 1
    turn[0] += 1
    with torch.cuda.stream(side[k]), torch.no_grad():
        c = [cams[turn[0] % 8]]
        strategies, tasks = start_strategy_final(c, hist)
        pkg = distributed_preprocess3dgs_and_all2all_final(c, model, pipe, bg, batched_strategies=strategies, mode="test")
        images, _ = render_final(pkg, strategies, late=pending[k])
    settle(pending[k ^ 1])
    return images


n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for _ in range(20):
    view()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    view()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"views/s {n / (time.perf_counter() - t0):.1f}; host {1e3 * host / n:.4f} ms per view")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    view()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(r"grendel-gs_amd|view_loop", 30)
print(s.getvalue()[:9000])
