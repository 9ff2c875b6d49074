"""Process-global state read by the hot path's host side.

Stand-alone stand-in for the slice of the reference's utils/general_utils.py that the
`gaussian_renderer` mirror touches (names and meaning identical to utils/general_utils.py:24-113,
161-169,194-269 there), so that bench.py / tests / smoke run on a box without the reference tree.
When the mirror is dropped into the reference tree, `import utils.general_utils` resolves to the
reference's own module instead and nothing here is used.
"""
import os
from types import SimpleNamespace

import torch
import torch.distributed as dist

ARGS = None
LOG_FILE = None
CUR_ITER = 0
GLOBAL_RANK = 0
LOCAL_RANK = 0
WORLD_SIZE = 1
DP_GROUP = None
MP_GROUP = None
DEFAULT_GROUP = None
IN_NODE_GROUP = None
TIMERS = None
DENSIFY_ITER = 0

BLOCK_X, BLOCK_Y = 16, 16
ONE_DIM_BLOCK_SIZE = 256
IMG_H, IMG_W = None, None
TILE_Y, TILE_X = None, None


def default_args(**overrides):
    """the flags the `final` path reads, with the reference's defaults (arguments/__init__.py:107-201)"""
    a = SimpleNamespace(
        bsz=1, log_interval=250, log_folder="/tmp/gs_log", zhx_debug=False, zhx_time=False,
        image_distribution=True, image_distribution_mode="final", gaussians_distribution=True,
        heuristic_decay=0.0, no_heuristics_update=False, border_divpos_coeff=1.0,
        adjust_strategy_warmp_iterations=-1, local_sampling=False, distributed_dataset_storage=False,
        lambda_dssim=0.2, lr_scale_loss=1.0, backend="default", save_strategy_history=False,
        redistribute_gaussians_mode="random_redistribute", redistribute_gaussians_frequency=10,
        redistribute_gaussians_threshold=1.1, sync_grad_mode="dense",  # arguments/__init__.py:148-156
    )
    for k, v in overrides.items():
        setattr(a, k, v)
    return a


def set_args(args):
    global ARGS
    ARGS = args


def get_args():
    return ARGS


def set_log_file(f):
    global LOG_FILE
    LOG_FILE = f


def get_log_file():
    return LOG_FILE


def set_cur_iter(i):
    global CUR_ITER
    CUR_ITER = i


def get_cur_iter():
    return CUR_ITER


def set_timers(t):
    global TIMERS
    TIMERS = t


def get_timers():
    return TIMERS


def set_block_size(x, y, z):
    global BLOCK_X, BLOCK_Y, ONE_DIM_BLOCK_SIZE
    BLOCK_X, BLOCK_Y, ONE_DIM_BLOCK_SIZE = x, y, z


def set_img_size(h, w):
    global IMG_H, IMG_W, TILE_Y, TILE_X
    IMG_H, IMG_W = h, w
    TILE_Y = (IMG_H + BLOCK_Y - 1) // BLOCK_Y
    TILE_X = (IMG_W + BLOCK_X - 1) // BLOCK_X


def get_img_size():
    return IMG_H, IMG_W


def get_img_width():
    return IMG_W


def get_img_height():
    return IMG_H


def get_num_pixels():
    return IMG_H * IMG_W


def get_denfify_iter():  # (sic) utils/general_utils.py:116
    return DENSIFY_ITER


def inc_densify_iter():
    global DENSIFY_ITER
    DENSIFY_ITER += 1


def check_initial_gpu_memory_usage(prefix):
    return None


class SingleGPUGroup:
    def rank(self):
        return 0

    def size(self):
        return 1


def one_node_device_count():
    n = torch.cuda.device_count() if torch.cuda.is_available() else WORLD_SIZE
    return min(max(n, 1), WORLD_SIZE)


def get_first_rank_on_cur_node():
    n = one_node_device_count()
    return (GLOBAL_RANK // n) * n


def init_distributed(args=None, backend=None):
    """one process per GPU; "nccl" IS RCCL on ROCm.  `backend="gloo"` is used by the CPU tests."""
    global GLOBAL_RANK, LOCAL_RANK, WORLD_SIZE, DEFAULT_GROUP, IN_NODE_GROUP
    GLOBAL_RANK = int(os.environ.get("RANK", 0))
    LOCAL_RANK = int(os.environ.get("LOCAL_RANK", 0))
    WORLD_SIZE = int(os.environ.get("WORLD_SIZE", 1))
    if WORLD_SIZE > 1:
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(LOCAL_RANK)
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=GLOBAL_RANK, world_size=WORLD_SIZE)
        DEFAULT_GROUP = dist.group.WORLD
        IN_NODE_GROUP = dist.group.WORLD  # single node (the driver never launches more)
    else:
        DEFAULT_GROUP = SingleGPUGroup()
        IN_NODE_GROUP = SingleGPUGroup()


def device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def our_allgather_among_cpu_processes_float_list(data, group):
    assert isinstance(data, list) and isinstance(data[0], float), "data should be a list of float"
    if group.size() == 1:
        return [list(data)]  # nothing to gather: no device round trip (the reference pays a full sync here)
    t = torch.tensor(data, dtype=torch.float32, device=device())
    if group.size() > 1:
        out = torch.empty((group.size() * len(data),), dtype=torch.float32, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        out = out.view(group.size(), len(data))
    else:
        out = t.unsqueeze(0)
    return out.cpu().tolist()
