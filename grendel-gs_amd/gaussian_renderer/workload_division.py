"""Pixel-partition API of the hot path (host side), MI355X build.

From-scratch mirror of the `*_final` surface of the reference's gaussian_renderer/workload_division.py
(names, arguments, return values and cut-point arithmetic identical; the dead non-final classes,
SURVEY.md F4, are not provided):

  division_pos_heuristic          workload_division.py:75-94
  DivisionStrategyFinal           workload_division.py:684-803
  DivisionStrategyHistoryFinal    workload_division.py:806-849
  start_strategy_final            workload_division.py:852-941
  finish_strategy_final           workload_division.py:944-998

A batch of B cameras is a list of B*TILE_Y tile ROWS; it is cut into WORLD_SIZE contiguous parts of
equal estimated cost, so with B >= W whole images land on single GPUs and with B < W an image is split
into row bands.  Difference from the reference in HOW (not WHAT): the per-row cost heuristics live on the
host (they are TILE_Y floats per camera), so choosing the cut points costs no device sync per iteration
(the reference's cumsum/searchsorted run on the GPU and read back, workload_division.py:92).
"""
import numpy as np
import torch

import diff_gaussian_rasterization
import utils.general_utils as utils


def get_tile_pixel_range(j, i, image_width, image_height):
    """pixel rect [minx,maxx) x [miny,maxy) of tile (row j, column i)"""
    x0, y0 = i * utils.BLOCK_X, j * utils.BLOCK_Y
    return x0, y0, min(x0 + utils.BLOCK_X, image_width), min(y0 + utils.BLOCK_Y, image_height)


def get_tile_pixel_cnt(j, i, image_width, image_height):
    x0, y0, x1, y1 = get_tile_pixel_range(j, i, image_width, image_height)
    return (x1 - x0) * (y1 - y0)


def division_pos_heuristic(heuristic, tile_num, world_size, right=False):
    """cut `tile_num` rows with per-row cost `heuristic` into `world_size` parts of equal cost:
    [0, searchsorted(cumsum(h), k * total / W, right) for k = 1..W-1, tile_num]"""
    assert heuristic.shape[0] == tile_num, "the length of heuristics should be the same as the number of tiles."
    # fp32 throughout, as the reference computes it (tests/test_gpu_reference_b1.py holds the cut points to the
    # reference's device cumsum / searchsorted on 400 seeded cost vectors); numpy: these are TILE_Y * bsz numbers
    h = heuristic.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(heuristic) else \
        np.asarray(heuristic, dtype=np.float32)
    prefix = np.cumsum(h, dtype=np.float32)
    per_worker = np.float32(prefix[-1] / np.float32(world_size))
    thresholds = np.arange(1, world_size, dtype=np.float32) * per_worker
    cuts = np.searchsorted(prefix, thresholds, side="right" if right else "left")
    return [0] + cuts.tolist() + [tile_num]


_MASK_CACHE = {}


def _device():
    """the current HIP device (not a `utils` helper: under graft level B2 `utils` is the reference's module)"""
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class DivisionStrategyFinal:
    """row-band partition of ONE camera among the `gpu_ids` that render a part of it"""

    def __init__(self, camera, world_size, gpu_ids, division_pos, gpu_for_this_camera_tilelr):
        assert world_size > 0, "The world_size must be greater than 0."
        assert len(gpu_ids) == world_size, "The number of gpu_ids must be equal to the world_size."
        assert len(division_pos) == world_size + 1, "The number of division_pos must be equal to the world_size+1."
        assert division_pos[0] == 0, "The first element of division_pos must be 0."
        assert division_pos[-1] == utils.TILE_Y, \
            "The last element of division_pos must be equal to the total number of tiles."
        assert all(b > a for a, b in zip(division_pos[:-1], division_pos[1:])), \
            "The division_pos must be in ascending order."
        for k, (l, r) in enumerate(gpu_for_this_camera_tilelr):
            assert l == division_pos[k] and r == division_pos[k + 1], \
                "The division_pos must be consistent with gpu_for_this_camera_tilelr."
        self.camera = camera
        self.world_size = world_size
        self.gpu_ids = gpu_ids
        self.rank = gpu_ids.index(utils.GLOBAL_RANK) if utils.GLOBAL_RANK in gpu_ids else -1
        self.division_pos = division_pos

    # -- which of my Gaussians does band j need (K2) -------------------------------------------
    def get_local2j_ids_bool(self, means2D, radii, image_height, image_width, cuda_args=None):
        div = torch.tensor(self.division_pos, dtype=torch.int32, device=means2D.device) * utils.TILE_X
        return diff_gaussian_rasterization._C.get_local2j_ids_bool(
            image_height, image_width, self.rank, self.world_size, means2D, radii, div, cuda_args)

    def get_local2j_ids(self, means2D, radii, raster_settings, cuda_args):
        """reference-compatible form: (list of [n,1] int64 index tensors, bool [P, ws]).  The fused
        exchange of this package uses get_local2j_ids_bool directly and never calls nonzero per band."""
        b = self.get_local2j_ids_bool(means2D, radii, raster_settings.image_height, raster_settings.image_width,
                                      cuda_args)
        return [b[:, j].nonzero() for j in range(self.world_size)], b

    # -- which tiles do I render ----------------------------------------------------------------
    def _my_rows(self):
        if utils.GLOBAL_RANK not in self.gpu_ids:
            return None
        k = self.gpu_ids.index(utils.GLOBAL_RANK)
        return self.division_pos[k], self.division_pos[k + 1]

    def get_compute_locally(self):
        rows = self._my_rows()
        if rows is None:
            return None
        # read-only masks, one per (band, grid, device), built once: no fill kernels in the steady state
        key = (rows[0], rows[1], utils.TILE_Y, utils.TILE_X, str(_device()))
        mask = _MASK_CACHE.get(key)
        if mask is None:
            if len(_MASK_CACHE) > 4096:
                _MASK_CACHE.clear()
            mask = torch.zeros((utils.TILE_Y, utils.TILE_X), dtype=torch.bool, device=_device())
            mask[rows[0]:rows[1]] = True
            _MASK_CACHE[key] = mask
        return mask

    def get_compute_locally_all(self):
        if self._my_rows() is None:
            return None
        return torch.ones((utils.TILE_Y, utils.TILE_X), dtype=torch.bool, device=_device())

    def get_extended_compute_locally(self):
        return None


class DivisionStrategyHistoryFinal:
    """per-camera row-cost heuristics + a log of measured times"""

    def __init__(self, dataset, world_size, rank):
        self.world_size = world_size
        self.rank = rank
        self.accum_heuristic = {cam.uid: torch.ones((utils.TILE_Y,), dtype=torch.float32) for cam in dataset.cameras}
        self.history = []

    def store_stats(self, batched_cameras, gpu_camera_running_time, batched_strategies):
        per_camera = [0.0] * len(batched_cameras)
        per_gpu = [0.0] * self.world_size
        infos = []
        for k, (camera, strategy) in enumerate(zip(batched_cameras, batched_strategies)):
            times = [gpu_camera_running_time[g][k] for g in strategy.gpu_ids]
            for g, t in zip(strategy.gpu_ids, times):
                per_camera[k] += t
                per_gpu[g] += t
            infos.append({"camera_id": camera.uid, "gpu_ids": strategy.gpu_ids,
                          "division_pos": strategy.division_pos, "each_gpu_running_time": times})
        self.history.append({"iteration": utils.get_cur_iter(), "all_gpu_running_time": per_gpu,
                             "all_camera_running_time": per_camera, "batched_camera_info": infos})

    def to_json(self):
        return self.history


def _snap_cuts_to_image_borders(cuts, rows_per_image, coeff):
    """a cut within `coeff` rows of an image boundary moves onto the boundary (saves one tiny band and
    its kernel launches; workload_division.py:889-902)"""
    for i in range(1, len(cuts) - 1):
        rem = cuts[i] % rows_per_image
        if rem + coeff >= rows_per_image:
            cuts[i] = (cuts[i] // rows_per_image + 1) * rows_per_image
        elif rem - coeff <= 0:
            cuts[i] = (cuts[i] // rows_per_image) * rows_per_image
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert a + coeff < b, "Each part between division_pos must be large enough."
    return cuts


def start_strategy_final(batched_cameras, strategy_history):
    """-> (batched_strategies [B], gpuid2tasks[gpu] = [(camera idx, row_l, row_r), ...])"""
    args = utils.get_args()
    W = utils.DEFAULT_GROUP.size()
    rows = utils.TILE_Y
    strategies = []
    gpuid2tasks = [[] for _ in range(W)]

    if args.local_sampling:
        per_gpu = args.bsz // utils.WORLD_SIZE
        for k, camera in enumerate(batched_cameras):
            g = k // per_gpu
            gpuid2tasks[g].append((k, 0, rows))
            strategies.append(DivisionStrategyFinal(camera, 1, [g], [0, rows], [(0, rows)]))
        return strategies, gpuid2tasks

    # the cut points are a function of the batch's row costs: a batch whose costs have not changed since it was last
    # planned (frozen heuristics, a converged balancer, bsz >= W) is not planned again (~50 us of host per iteration)
    hs = [strategy_history.accum_heuristic[c.uid] for c in batched_cameras]
    key = (W, rows, float(args.border_divpos_coeff)) + tuple((c.uid, id(h), h._version) for c, h in zip(batched_cameras, hs))
    cache = strategy_history.__dict__.setdefault("_gsr_cuts", {})
    hit = cache.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[1], hs)):  # (the tensors themselves: an id may be reused)
        cuts = list(hit[0])
    else:
        heur = torch.cat([h.to("cpu") for h in hs], dim=0)
        cuts = division_pos_heuristic(heur, rows * len(batched_cameras), W, right=True)
        cuts = _snap_cuts_to_image_borders(cuts, rows, args.border_divpos_coeff)
        if len(cache) > 8192:
            cache.clear()
        cache[key] = (tuple(cuts), tuple(hs))

    for k, camera in enumerate(batched_cameras):
        lo, hi = k * rows, (k + 1) * rows
        gpus, bands = [], []
        for g in range(W):
            gl, gr = cuts[g], cuts[g + 1]
            if gr <= lo or hi <= gl:
                continue
            band = (max(gl, lo) - lo, min(gr, hi) - lo)
            gpus.append(g)
            bands.append(band)
            gpuid2tasks[g].append((k, band[0], band[1]))
        strategies.append(DivisionStrategyFinal(camera, len(gpus), gpus, [0] + [b[1] for b in bands], bands))
    return strategies, gpuid2tasks


def _resolve_deferred_timings(st):
    """exact per-call milliseconds from the HIP event pairs the ops left behind (one wait on the last
    event instead of the reference's cuda.synchronize() pairs around every loss / render call)"""
    stamps = st.pop("_gsr_stamps", None)
    if stamps is not None:  # an iteration replayed as a hipGraph: device timestamps (graphed_step.GraphedIteration)
        st.update(stamps())
    for key, evkey in (("forward_render_time", "_fwd_events"), ("backward_render_time", "_bwd_events"),
                       ("forward_loss_time", "_loss_events")):
        ev = st.pop(evkey, None)
        if ev is not None:
            ev[1].synchronize()
            st[key] = float(ev[0].elapsed_time(ev[1]))


# How the load balancer gets its timings at W > 1 (bsz < W, large images: the heuristics are live):
#   "exact"     : the reference's semantics (workload_division.py:953-966) -- this step's render / loss times, which costs
#                 a wait for this step's backward plus a device all-gather + read-back: the host cannot run ahead;
#   "pipelined" : the PREVIOUS step's times (their HIP events completed long ago: resolving them waits for nothing),
#                 all-gathered over a host-side (gloo) group -- no device synchronisation at all, the cut points lag one
#                 step behind (SURVEY.md 7 lists this as the documented alternative).
# "exact" is the default (the reference's schedule: an iteration's cut points follow from the iteration before it);
# "pipelined" is opt-in (bench.py --balance-timing pipelined) and changes WHEN a measurement takes effect, not how.
_BALANCE = {"mode": "exact", "group": None}


def set_balance_timing(mode):
    assert mode in ("exact", "pipelined")
    _BALANCE["mode"] = mode
    if mode == "pipelined":
        _host_group()  # a collective (dist.new_group): made here, by every rank, not inside a training step


def flush_balance_timing(strategy_history):
    """"pipelined" holds the last iteration's timings back until the next call of finish_strategy_final; call this at
    the end of training / before saving the strategy history so that the final iteration is logged too"""
    prev = getattr(strategy_history, "_gsr_pending", None)
    if prev is None:
        return
    strategy_history._gsr_pending = None
    p_cams, p_strategies, p_stats, p_frozen = prev
    for st in p_stats:
        _resolve_deferred_timings(st)
    times = _gather_times_on_host(_my_times(p_strategies, p_stats))
    _update_heuristics(p_cams, strategy_history, p_strategies, times, p_frozen)


def _host_group():
    """a gloo group over the same ranks: the timing all-gather is a host-only operation.  Returns None when gloo
    cannot be brought up (the caller then gathers through the device group, as the reference does)"""
    import os

    import torch.distributed as dist

    if _BALANCE["group"] is None:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        lws, ws = os.environ.get("LOCAL_WORLD_SIZE"), os.environ.get("WORLD_SIZE")
        if lws is not None and ws is not None and lws == ws:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one torchrun node: the hostname may not resolve
        try:
            _BALANCE["group"] = dist.new_group(backend="gloo")  # collective: every rank makes its first call here
        except Exception as e:  # noqa: BLE001
            print(f"[workload_division] no host-side gloo group ({type(e).__name__}: {e}); timing gather stays on the "
                  f"device group", flush=True)
            _BALANCE["group"] = False
    return _BALANCE["group"] or None


def _gather_times_on_host(mine):
    import torch.distributed as dist

    g = _host_group()
    if g is None:
        return utils.our_allgather_among_cpu_processes_float_list(mine, utils.DEFAULT_GROUP)
    # (a tensor all-gather: all_gather_object pickles and runs two collectives, several times the cost for W short lists)
    t = torch.tensor([float(x) for x in mine], dtype=torch.float64)
    outs = [torch.empty_like(t) for _ in range(g.size())]
    dist.all_gather(outs, t, group=g)
    return [o.tolist() for o in outs]


def _update_heuristics(batched_cameras, strategy_history, batched_strategies, times, frozen):
    args = utils.get_args()
    strategy_history.store_stats(batched_cameras, times, batched_strategies)
    if frozen:
        return
    for k, (camera, strategy) in enumerate(zip(batched_cameras, batched_strategies)):
        new = torch.zeros((utils.TILE_Y,), dtype=torch.float32)
        for j, g in enumerate(strategy.gpu_ids):
            l, r = strategy.division_pos[j], strategy.division_pos[j + 1]
            new[l:r] = times[g][k] / (r - l)
        if args.heuristic_decay == 0:
            strategy_history.accum_heuristic[camera.uid] = new
        else:
            old = strategy_history.accum_heuristic[camera.uid].to("cpu")
            strategy_history.accum_heuristic[camera.uid] = old * args.heuristic_decay + new * (1 - args.heuristic_decay)


def _my_times(batched_strategies, batched_statistic_collector):
    mine = []
    for k, strategy in enumerate(batched_strategies):
        if utils.GLOBAL_RANK not in strategy.gpu_ids:
            mine.append(-1.0)
            continue
        st = batched_statistic_collector[k]
        mine.append(float(st["forward_render_time"] + st["backward_render_time"] + st["forward_loss_time"] * 2))
    return mine


def timings_have_consumer():
    """Does anybody read this iteration's render / loss timings?  The load balancer does when its heuristics are live
    (the skip rules of workload_division.py:968-978), the strategy history does when it is saved
    (train_internal.py:274-284).  Otherwise -- always at world size 1 -- the ops record no HIP events at all (six event
    packets per camera are ~2 % of a 1.3 ms iteration) and the statistics keep their 0.0 placeholders."""
    args = utils.get_args()
    W = utils.DEFAULT_GROUP.size()
    small = utils.get_img_height() <= 600 or utils.get_img_width() <= 1000
    whole_images = args.bsz >= W and (utils.get_img_height() <= 1080 or utils.get_img_width() <= 1920)
    frozen = (utils.get_cur_iter() <= args.adjust_strategy_warmp_iterations or W == 1 or args.no_heuristics_update
              or whole_images or small)
    return (not frozen) or bool(getattr(args, "save_strategy_history", False))


def finish_strategy_final(batched_cameras, strategy_history, batched_strategies, batched_statistic_collector):
    """all-gather each rank's measured (fwd render + bwd render + 2 x fwd loss) ms per camera and turn
    it into the next per-row cost estimate (same skip rules as workload_division.py:968-978)"""
    args = utils.get_args()
    W = utils.DEFAULT_GROUP.size()
    small = utils.get_img_height() <= 600 or utils.get_img_width() <= 1000
    whole_images = args.bsz >= W and (utils.get_img_height() <= 1080 or utils.get_img_width() <= 1920)
    frozen = (utils.get_cur_iter() <= args.adjust_strategy_warmp_iterations or W == 1 or args.no_heuristics_update
              or whole_images or small)
    if frozen and W > 1 and not getattr(args, "save_strategy_history", False):
        # nobody consumes the timings (heuristics frozen, history not saved: train_internal.py:274-284): skip the
        # event wait and the all-gather + host read-back the reference pays every iteration (:953-966), so the
        # host keeps running ahead of the device.  The HIP event pairs are simply dropped.
        for st in batched_statistic_collector:
            for evkey in ("_fwd_events", "_bwd_events", "_loss_events", "_gsr_stamps"):
                st.pop(evkey, None)
        return

    if W > 1 and _BALANCE["mode"] == "pipelined":
        prev = getattr(strategy_history, "_gsr_pending", None)
        strategy_history._gsr_pending = (batched_cameras, batched_strategies, batched_statistic_collector, frozen)
        if prev is None:
            return  # first step: nothing measured yet
        p_cams, p_strategies, p_stats, p_frozen = prev
        for st in p_stats:
            _resolve_deferred_timings(st)  # events of the previous step: complete, no wait
        times = _gather_times_on_host(_my_times(p_strategies, p_stats))
        _update_heuristics(p_cams, strategy_history, p_strategies, times, p_frozen)
        return

    if W > 1:
        for st in batched_statistic_collector:
            _resolve_deferred_timings(st)
    mine = _my_times(batched_strategies, batched_statistic_collector)
    if W == 1:
        times = [list(mine)]  # nothing to gather: no device round trip (the reference's helper syncs the device here)
    else:
        times = utils.our_allgather_among_cpu_processes_float_list(mine, utils.DEFAULT_GROUP)
    _update_heuristics(batched_cameras, strategy_history, batched_strategies, times, frozen)
