"""GraphedIteration -- one training iteration of the hot path replayed as ONE hipGraph.

Why: behind the pixel partition a rank's iteration is ~50 kernels of 3-200 us (0.8 ms in all at 8 ranks on
BASELINE configs[2]'s shape) and the Python / autograd work between two C-ABI calls costs more than the kernels do --
the W > 1 step is HOST-bound (profiles/r03s2_fake_world_w8_gaps.txt: 0.94 ms of kernels, 0.7-0.95 ms of launch gaps).
The reference's loop is eager too (train_internal.py:134-208, 316-329), but it is not sub-millisecond.  Capturing the
iteration -- GT band staging, K1, the exchange (count, pack into capacity slabs, all-to-all-v, unpack), K3-K8, the band
loss, K10, the mirror exchange, the fused K11 + Adam launch -- into a hipGraph leaves ONE launch per iteration on the host.

What made the iteration capturable (every item is a host dependency the eager path has):
  * pair count D (the reference reads `num_rendered` back to size its sort): the tile sort runs at a CAPACITY taken
    from earlier iterations (gsr_bin_sort_bounded), D stays on the device; D > capacity raises a bit of a device flag word;
  * exchange sizes (gaussian_renderer/__init__.py:572-585 reads the i2j counts back): capacity slabs from the planner;
    gsr_exchange_check raises the flag when a count exceeds its slab or a rendered band receives < 10 rows;
  * Adam's step count / learning rates (kernel arguments in the eager launch): read from a 12-float device block that
    the host refreshes with an asynchronous copy in front of every replay (gsr_preprocess_backward_adam_raw_batched_dyn);
  * per-iteration inputs (camera, ground truth): static proxies refreshed by device copies in front of the replay.
A raised flag makes the optimizer launch of that replay -- and, the flag being sticky, of every later one -- a no-op; the
host looks at the flag one iteration late (so it never waits for the replay it has just launched), and on a raised flag
repeats the affected iterations EAGERLY, in order: results are those of the eager loop, iteration by iteration.

One graph per (image size, camera intrinsics, shard size, capacities) -- and NOT per partition (round 6): with more
than one rank the row bands are device data too.  The captured launches are sized for a band CAPACITY (the tallest band
this rank has been given, plus slack) and read the band itself at execution time: the exchange from the [B,W,2] table it
always read, compute_locally from a launch that rebuilds the masks out of the same block (gsr_band_mask), K8 / K10 from
the row hull the tile sort leaves behind the range table (row_lo = -1), the loss kernels from { y0, y1 }
(gsr_l1_ssim_*_band); the ground-truth band lands in the first rows of a capacity-sized buffer.  The host only checks
that a band fits the capacity.  So every camera of a live partition -- the reference keeps cut points PER CAMERA and
moves them with the measured times (workload_division.py:806-849) -- replays the same graph; only a change of the SET of
ranks that render a camera (bsz > 1) is a new key.  A key's first `warmup` sights run eagerly (they teach the
capacities); a band taller than the captured capacity, or an exchange slab the planner wants larger than the captured
layout (the planner's capacities plus an eighth), replaces the graph after ONE eager iteration (_usable).
timings=True: the replays carry device timestamps around K3-K8, the loss forward and K10 (gsr_stamp), which
`last_stats` hands to finish_strategy_final in place of the eager ops' HIP events: the load balancer runs on replayed
iterations (tests/test_gpu_dynamic_bands.py: 157 of 160 iterations replayed from 2 captures while the cut points moved
127 times).  Without timings=True the wrapper is for loops in which nothing consumes per-iteration timings (frozen
heuristics): events recorded inside a capture cannot be read.

    step = GraphedIteration(optimizer, body)      # body(cameras, strategies, tasks) -> loss: GT staging ... opt.step()
    loss = step(cameras, strategies, tasks)       # replays when it can, runs `body` eagerly when it cannot
    step.validate()                               # before READING results (loss.item(), means2D.grad, ...)
"""
import copy
import math
import os as _os
import sys as _sys
from types import SimpleNamespace

import torch

import diff_gaussian_rasterization as _dgr

_DEBUG = _os.environ.get("GSR_GRAPH_DEBUG") == "1"  # every replay synchronised and logged to stderr


def _dbg(*a):
    if _DEBUG:
        print("[graphed]", *a, file=_sys.stderr, flush=True)


class _Entry:
    __slots__ = ("graph", "proxies", "out", "ctx", "caps_key", "pair_cap", "sproxies", "tasks", "band_cap", "masks")


class _BandProxy:
    """stands in for a DivisionStrategyFinal while an iteration is captured for ANY band of at most `cap` tile rows: the
    band's rows are device words the replay refreshes.  Reading division_pos on the host would bake one partition into
    the graph, so it raises (a capture that trips over it fails, and the iteration stays eager)."""

    def __init__(self, real, cap, mask, band_rows, all_bands):
        import utils.general_utils as utils

        self.camera, self.world_size, self.gpu_ids, self.rank = real.camera, real.world_size, list(real.gpu_ids), real.rank
        self._renders = utils.GLOBAL_RANK in self.gpu_ids
        self._cap, self._mask = int(cap), mask
        self._gsr_dyn_band = band_rows        # int32 [2] on the device: { y0, y1 } pixel rows of this rank's band
        self._gsr_dyn_all_bands = all_bands   # int32 [B,W,2] on the device: tile rows of every rank and camera

    @property
    def division_pos(self):
        raise RuntimeError("band-agnostic graph capture: division_pos is device data here (graphed_step._BandProxy)")

    def _my_rows(self):
        return (-1, self._cap) if self._renders else None  # (include/gsraster.h: row_lo == -1)

    def get_compute_locally(self):
        return self._mask if self._renders else None

    def get_compute_locally_all(self):
        raise RuntimeError("band-agnostic graph capture: get_compute_locally_all is not captured")

    def get_extended_compute_locally(self):
        return None

    def get_local2j_ids_bool(self, *a, **k):
        raise RuntimeError("band-agnostic graph capture: the exchange reads the band table, not a strategy")

    get_local2j_ids = get_local2j_ids_bool


class GraphedIteration:
    def __init__(self, optimizer, body, warmup=3, max_graphs=8, enabled=True, dynamic_bands=True, timings=False):
        self.opt, self.body, self.warmup, self.max_graphs = optimizer, body, int(warmup), int(max_graphs)
        self.enabled = bool(enabled)
        # False (or GSR_GRAPH_DYNAMIC_BANDS=0, for A/B measurements): one graph per partition (rounds 3-5)
        self.dynamic_bands = bool(dynamic_bands) and _os.environ.get("GSR_GRAPH_DYNAMIC_BANDS", "1") != "0"
        self._band_cap = 0                        # tile rows the band-agnostic launches are sized for
        # timings: the replays carry device timestamps around K3-K8, the loss forward and K10 of every camera (six
        # one-thread launches per camera); `last_stats` is then, after a replay, what the eager ops leave in their
        # stats_collector dicts -- per camera { forward_render_time, backward_render_time, forward_loss_time } in ms,
        # resolved when somebody reads them (workload_division._resolve_deferred_timings) -- and None after an iteration
        # that ran eagerly (the body's own stats carry HIP events then).  This is what lets the load balancer
        # (finish_strategy_final) run on replayed iterations.
        self.timings = bool(timings)
        self.last_stats = None
        self.entries, self._seen = {}, {}
        self._inflight = None       # (event, cameras, strategies, tasks) of the replay nobody has validated yet
        self._flag = self._dyn = None
        self._hyper_host = None
        self.stats = {"replayed": 0, "eager": 0, "captured": 0, "redone": 0, "disabled": None}

    # ------------------------------------------------------------------ keys and static inputs
    def _planner_caps(self, B):
        """(planner, capacities int64 numpy [W,W,B]) of the exchange this rank takes part in, or (None, None)"""
        import gaussian_renderer as gr
        import torch.distributed as dist
        import utils.general_utils as utils

        group = utils.DEFAULT_GROUP
        if group.size() == 1:
            if not (gr._EXCHANGE_OPTIONS["forced"] and dist.is_initialized()):
                return None, None
            if not isinstance(group, dist.ProcessGroup):
                group = dist.group.WORLD
        planner = gr._planner(group, group.size(), B)
        return planner, planner.caps

    def _dynamic(self, strategies):
        """bands as device data: whenever a camera is split among ranks"""
        return self.dynamic_bands and any(len(s.gpu_ids) > 1 for s in strategies)

    @staticmethod
    def _my_bands(strategies):
        """[(lo, hi) tile rows of this rank's band, or None] per camera"""
        import utils.general_utils as utils

        out = []
        for s in strategies:
            if utils.GLOBAL_RANK in s.gpu_ids:
                j = s.gpu_ids.index(utils.GLOBAL_RANK)
                out.append((int(s.division_pos[j]), int(s.division_pos[j + 1])))
            else:
                out.append(None)
        return out

    def _key(self, cameras, strategies):
        import utils.general_utils as utils

        p0 = self.opt.param_groups[0]["params"][0]
        if self._dynamic(strategies):
            part = ("bands",) + tuple(tuple(s.gpu_ids) for s in strategies)
        else:
            part = tuple((tuple(s.gpu_ids), tuple(s.division_pos)) for s in strategies)
        # (capacities -- the band's, the exchange slabs' -- are properties of the ENTRY, see _usable: a capacity that
        # no longer holds replaces the graph after one eager iteration, without a new warm-up)
        return (part, utils.get_img_size(),
                tuple((float(c.FoVx), float(c.FoVy), int(c.image_height), int(c.image_width)) for c in cameras),
                p0.data_ptr(), tuple(p0.shape))

    @staticmethod
    def _band_rows(strategies):
        """the tallest band of the batch on ANY rank: replay-or-eager has to be ONE decision of the whole group (an eager
        rank and a replaying one would run different slab layouts against each other), so it may only depend on what
        every rank knows -- the partition -- and never on this rank's own band"""
        return max([int(b) - int(a) for s in strategies for a, b in zip(s.division_pos[:-1], s.division_pos[1:])] or [0])

    def _usable(self, entry, cameras, strategies):
        """do the capacities this graph was captured with still hold -- decided on the host BEFORE the replay, from
        numbers every rank has (the partition, the exchange planner's capacities): the tallest band of this batch (on any rank) fits
        the launches' band capacity, and the planner asks for no slab larger than the captured layout's"""
        if entry.sproxies is not None and self._band_rows(strategies) > entry.band_cap:
            return False
        _, caps = self._planner_caps(len(cameras))
        if (caps is None) != (entry.caps_key is None):
            return False
        return caps is None or (caps.shape == entry.caps_key.shape and bool((caps <= entry.caps_key).all()))

    @staticmethod
    def _packed(camera):
        """the [40] record K1 / K11 read (diff_gaussian_rasterization.pack_camera), cached on the camera under the key
        the mirror checks (gaussian_renderer/__init__.py: distributed_preprocess3dgs_and_all2all_final)"""
        tf = (math.tan(camera.FoVx * 0.5), math.tan(camera.FoVy * 0.5))
        mats = (camera.world_view_transform, camera.full_proj_transform, camera.camera_center)
        key = (float(tf[0]), float(tf[1])) + tuple((t.data_ptr(), t._version) for t in mats)
        cached = getattr(camera, "_gsr_packed", None)
        if cached is None or cached[0] != key:
            rs = SimpleNamespace(viewmatrix=mats[0], projmatrix=mats[1], campos=mats[2], tanfovx=tf[0], tanfovy=tf[1])
            cached = (key, _dgr.pack_camera(rs))
            camera._gsr_packed = cached
        return cached[1]

    MAXB = 16  # cameras per batch whose packed records live in the device block (larger batches: not graphed)

    @staticmethod
    def _packed_host(camera):
        """numpy copy of the camera's [40] record (one device read-back per camera, ever: cameras are static)"""
        dev = GraphedIteration._packed(camera)
        key = camera._gsr_packed[0]
        cached = getattr(camera, "_gsr_packed_host", None)
        if cached is None or cached[0] != key:
            cached = (key, dev.detach().cpu().numpy().copy())
            camera._gsr_packed_host = cached
        return cached[1]

    def _make_proxies(self, cameras, tasks):
        """static stand-ins of the batch's cameras: the captured kernels read THEIR buffers -- the [40] camera record
        (a slice of the device block that also carries Adam's constants: ONE host-to-device copy refreshes both) and
        the uint8 ground-truth band of the rows this rank renders (refreshed by one strided copy per camera)"""
        import utils.general_utils as utils
        from gaussian_renderer.loss_distribution import get_coverage_y_min_max

        mine = {k: (l, r) for (k, l, r) in tasks[utils.GLOBAL_RANK]} if tasks is not None else {}
        proxies = []
        for k, cam in enumerate(cameras):
            px = copy.copy(cam)
            for name in ("world_view_transform", "full_proj_transform", "camera_center"):
                setattr(px, name, getattr(cam, name).detach().clone())
            mats = (px.world_view_transform, px.full_proj_transform, px.camera_center)
            key = (float(math.tan(px.FoVx * 0.5)), float(math.tan(px.FoVy * 0.5))) + tuple(
                (t.data_ptr(), t._version) for t in mats)
            px._gsr_packed = (key, self._dyn[16 + 40 * k:16 + 40 * (k + 1)])
            # the ground truth: only the band this rank renders, in a static buffer that the mirror's band cache
            # (gaussian_renderer/loss_distribution.py: _band_of) finds under the placeholder's key -- no copy is captured
            px.original_image = None
            px._gsr_band = None
            dev = self._dyn.device
            px.original_image_backup = torch.empty((3, 1, 1), dtype=torch.uint8, device=dev)  # placeholder (cache key)
            px._gsr_bands = []
            if k in mine:
                y0, y1 = get_coverage_y_min_max(*mine[k])
                band = torch.empty((3, y1 - y0, int(cam.image_width)), dtype=torch.uint8, device=dev)
                src = px.original_image_backup
                px._gsr_bands = [((y0, y1, src.data_ptr(), src._version), band)]
                px._gsr_band = (y0, y1, band)
            proxies.append(px)
        return proxies

    def _strategy_proxies(self, entry, strategies, dev):
        """band-agnostic capture: stand-ins of the strategies (-> (proxies, tasks for the ground-truth staging)) whose
        masks one captured launch rebuilds from the band words of the device block"""
        import utils.general_utils as utils

        B, W, cap = len(strategies), self._world, entry.band_cap
        i32 = self._dyn.view(torch.int32)
        gy, gx = int(utils.TILE_Y), int(utils.TILE_X)
        entry.masks = torch.zeros((B, gy, gx), dtype=torch.uint8, device=dev)
        all_bands = i32[self.ALLB:self.ALLB + 2 * W * B].view(B, W, 2)
        sproxies = [_BandProxy(s, cap, entry.masks[k].view(torch.bool),
                               i32[self.BANDS + 4 * k + 2:self.BANDS + 4 * k + 4], all_bands)
                    for k, s in enumerate(strategies)]
        tasks = [[] for _ in range(max(W, utils.GLOBAL_RANK + 1))]
        tasks[utils.GLOBAL_RANK] = [(k, 0, cap) for k, s in enumerate(strategies) if utils.GLOBAL_RANK in s.gpu_ids]
        return sproxies, tasks

    def _refresh(self, entry, cameras, strategies, slot):
        """stage this iteration's inputs: camera records -- and, for a band-agnostic graph, the bands of every rank --
        into the pinned block of `slot` (they travel with Adam's constants), ground-truth bands by one strided copy each"""
        from gaussian_renderer.loss_distribution import get_coverage_y_min_max

        proxies = entry.proxies
        mine = self._my_bands(strategies) if entry.sproxies is not None else None
        if mine is not None:
            W = self._world
            words = self._hyper_seq[slot]
            for k, s in enumerate(strategies):
                lo, hi = mine[k] or (0, 0)
                y0, y1 = get_coverage_y_min_max(lo, hi) if mine[k] else (0, 0)
                words[self.BANDS + 4 * k:self.BANDS + 4 * k + 4] = (lo, hi, y0, y1)
                row = words[self.ALLB + 2 * W * k:self.ALLB + 2 * W * (k + 1)]
                row[:] = 0
                for j, g in enumerate(s.gpu_ids):
                    row[2 * g], row[2 * g + 1] = s.division_pos[j], s.division_pos[j + 1]
        for k, (px, cam) in enumerate(zip(proxies, cameras)):
            self._hyper_np[slot, 16 + 40 * k:16 + 40 * (k + 1)] = self._packed_host(cam)
            if px._gsr_band is not None:
                y0, y1, band = px._gsr_band
                if mine is not None:  # the band of THIS iteration, in the first rows of the capacity-sized buffer
                    y0, y1 = get_coverage_y_min_max(*mine[k])
                    band = band[:, :y1 - y0, :]
                band.copy_(cam.original_image_backup[:, y0:y1, :], non_blocking=True)
            px.uid = getattr(cam, "uid", None)

    # ------------------------------------------------------------------ capture
    RING = 16  # replays whose result slots / hyper-parameter blocks may be outstanding (two ever are)
    STAMPS = 6 * MAXB  # device timestamps per replay

    def _ensure_buffers(self, dev):
        if self._flag is None:
            self._flag = torch.zeros((1,), dtype=torch.int32, device=dev)
            # device block the captured launches read at execution time: 12 Adam constants, the replay's sequence number
            # (word 12) and the batch's camera records (40 floats each, from word 16)
            # ... and, from word BANDS, the row bands (int32): { lo, hi, y0, y1 } of this rank per camera, then from word
            # ALLB the [B,W,2] table of every rank's tile rows that the exchange reads
            import utils.general_utils as utils

            self._world = W = max(int(utils.DEFAULT_GROUP.size()), 1)
            self.BANDS = 16 + 40 * self.MAXB
            self.ALLB = self.BANDS + 4 * self.MAXB
            words = self.ALLB + 2 * W * self.MAXB
            self._dyn = torch.zeros((words,), dtype=torch.float32, device=dev)
            self._hyper_host = torch.zeros((self.RING, words), dtype=torch.float32).pin_memory()
            self._hyper_np = self._hyper_host.numpy()
            self._hyper_seq = self._hyper_host.view(torch.int32).numpy()
            # pinned, device-mapped ring the LAST launch of every replay stores { flag, sequence number } into
            self._ring = torch.zeros((2 * self.RING,), dtype=torch.int32).pin_memory()
            self._ring_np = self._ring.numpy()
            # ... and the device timestamps of a replay (timings=True): STAMPS words per slot
            self._stamp_ring = torch.zeros((self.RING * self.STAMPS,), dtype=torch.int64).pin_memory()
            self._stamp_np = self._stamp_ring.numpy()
            self._seq = 0

    def _capture(self, key, cameras, strategies, tasks):
        if _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0" and _os.environ.get("GSR_GRAPH_ANYWAY") != "1":
            # Measured on ROCm 7.0.2 (tools/probes/graph_bench_probe2.py): with the runtime's graph packet capture on, a
            # graph's pre-built packets go stale once a few hundred ordinary launches have run between two replays
            # (400 trivial elementwise launches are enough) and the next replay dies with a memory access fault.  With
            # the flag off replays cost ~3 % more and survive.  The runtime reads the flag when libamdhip64 is loaded:
            # it has to be in the environment before torch is imported (bench.py, tests/conftest.py set it).
            raise RuntimeError("hipGraph replays need DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE torch is "
                               "imported (HIP runtime issue with replays after eager launches); iteration stays eager")
        dev = self.opt.param_groups[0]["params"][0].device
        self._ensure_buffers(dev)
        # (the six (group, parameter) pairs of the fused launch are re-discovered by this capture: after a densification
        # event the ones of the previous capture name parameters that no longer exist)
        if self.opt.__dict__.get("_graph_owners") is not None:
            self.opt._graph_owners = None
        e = _Entry()
        if len(cameras) > self.MAXB:
            raise RuntimeError(f"batches above {self.MAXB} cameras are not graphed")
        import utils.general_utils as utils

        if getattr(utils.get_args(), "distributed_dataset_storage", False) and utils.DEFAULT_GROUP.size() > 1:
            raise RuntimeError("distributed dataset storage stages its ground truth with point-to-point sends: not graphed")
        e.sproxies = e.tasks = e.masks = None
        if self._dynamic(strategies):
            import utils.general_utils as utils

            # the tallest band any rank has been given, plus slack (idle workgroups are cheap, a new capture is not)
            rows = self._band_rows(strategies)
            self._band_cap = max(self._band_cap, min(int(utils.TILE_Y), rows + max(2, rows // 2)))
        e.band_cap = self._band_cap
        body_strategies, body_tasks = strategies, tasks
        if self._dynamic(strategies):
            e.sproxies, e.tasks = body_strategies, body_tasks = self._strategy_proxies(e, strategies, dev)
        e.proxies = self._make_proxies(cameras, body_tasks)
        self._stage(e, cameras, strategies)
        e.ctx = _dgr.GraphCapture(self._flag, self._dyn)
        if self.timings:
            e.ctx.stamp_ring = (self._stamp_ring, self.RING, self.STAMPS)
        planner, caps = self._planner_caps(len(cameras))
        e.caps_key = None
        if caps is not None:
            # the graph's own slab layout: the planner's capacities plus an eighth -- a layout is baked into the captured
            # all-to-all, and while the partition still moves every slab that outgrows it would cost a capture (the
            # eager path keeps the planner's own capacities: both layouts are self-consistent, and every rank derives
            # both from the same all-gathered history)
            slack = float(_os.environ.get("GSR_GRAPH_SLAB_SLACK", "1.125"))  # (1.0: the planner's own layout)
            e.caps_key = ((caps * slack).astype(caps.dtype) + 255) // 256 * 256 if slack != 1.0 else caps.copy()
            e.ctx.slab_caps_dev = torch.tensor(e.caps_key.reshape(-1), dtype=torch.int32).to(dev)
        torch.cuda.synchronize(dev)
        e.graph = torch.cuda.CUDAGraph()
        _dbg("capture begins")
        _dgr._CAPTURE[0] = e.ctx
        saved_caps = None
        if e.caps_key is not None:
            saved_caps = (planner.caps, planner.caps_list)
            planner.caps, planner.caps_list = e.caps_key, e.caps_key.tolist()
        try:
            # thread_local: the process group's watchdog thread keeps polling the events of EARLIER (eager) collectives
            # while this thread captures; in the default (global) mode HIP fails those polls and the watchdog aborts the
            # process (measured on RCCL 2.26 / ROCm 7.0)
            with torch.cuda.graph(e.graph, capture_error_mode="thread_local"):
                if e.sproxies is not None:  # compute_locally of every camera from the band words of this replay
                    _dgr.check(_dgr.lib.gsr_band_mask(int(e.masks.shape[2]), int(e.masks.shape[1]), len(cameras),
                                                      self._dyn.data_ptr() + 4 * self.BANDS, 4, e.masks.data_ptr(),
                                                      _dgr._stream()), "gsr_band_mask")
                e.out = self.body(e.proxies, body_strategies, body_tasks)
                _dgr.check(_dgr.lib.gsr_publish_flag(self._flag.data_ptr(), self._dyn.data_ptr() + 48,
                                                     self._ring.data_ptr(), self.RING, _dgr._stream()),
                           "gsr_publish_flag")
        finally:
            _dgr._CAPTURE[0] = None
            if saved_caps is not None:
                planner.caps, planner.caps_list = saved_caps
        if len(self.entries) >= self.max_graphs:
            self.entries.pop(next(iter(self.entries)))
        self.entries[key] = e
        self.stats["captured"] += 1
        _dbg("captured; pair capacities", [c for _, c in e.ctx.pairs])
        return e

    @staticmethod
    def _group_says_no(failed):
        """-> True when the capture failed on ANY rank of the default group (one eager 4-byte all-reduce)"""
        import torch.distributed as dist

        import utils.general_utils as utils

        group = utils.DEFAULT_GROUP
        if group is None or group.size() == 1 or not dist.is_initialized() or not isinstance(group, dist.ProcessGroup):
            return failed
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return bool(int(t.item()))

    def _stage(self, entry, cameras, strategies):
        """-> sequence number of the replay these inputs are staged for"""
        self._seq = seq = (self._seq + 1) & 0x3FFFFFFF or 1
        slot = seq % self.RING
        B = len(cameras)
        self._refresh(entry, cameras, strategies, slot)
        if self.opt.__dict__.get("_graph_owners") is not None:
            self._hyper_np[slot, :12] = self.opt.graph_hyper()
        self._hyper_seq[slot, 12] = seq
        n = 16 + 40 * B if entry.sproxies is None else self.ALLB + 2 * self._world * B
        self._dyn[:n].copy_(self._hyper_host[slot, :n], non_blocking=True)
        return seq

    # ------------------------------------------------------------------ timings of a replay
    def _replay_stats(self, entry, seq, ev, strategies):
        """-> per camera the dict an eager iteration's ops leave in stats_collector; the three times are filled in from the
        replay's device timestamps when "_gsr_stamps" is resolved (waits for the replay's publish stamp, nothing else)"""
        import utils.general_utils as utils

        zero = {"forward_render_time": 0.0, "backward_render_time": 0.0, "forward_loss_time": 0.0}
        if not self.timings or not entry.ctx.stamps:
            return [dict(zero) for _ in strategies]
        # stamps are tagged with the camera's stats_collector (an id): cameras in the order of their first stamp = the
        # cameras this rank renders, in batch order (render_final walks the batch in order)
        order, slots = [], {}
        for idx, (kind, tag) in enumerate(entry.ctx.stamps):
            if tag not in slots:
                slots[tag] = {}
                order.append(tag)
            slots[tag][kind] = idx
        rendered = [k for k, s in enumerate(strategies) if utils.GLOBAL_RANK in s.gpu_ids]
        out = [dict(zero) for _ in strategies]
        slot = seq % self.RING
        base = slot * self.STAMPS

        def resolver(idx_of):
            def resolve():
                spins = 0
                while int(self._ring_np[2 * slot + 1]) != seq:
                    spins += 1
                    if spins > 2000:
                        ev.synchronize()
                        break
                t = self._stamp_np

                def ms(a, b):
                    if a not in idx_of or b not in idx_of:
                        return 0.0
                    return max(float(int(t[base + idx_of[b]]) - int(t[base + idx_of[a]])) * 1e-5, 0.0)  # 100 MHz ticks

                return {"forward_render_time": ms("fwd0", "fwd1"), "backward_render_time": ms("bwd0", "bwd1"),
                        "forward_loss_time": ms("loss0", "loss1")}
            return resolve

        for k, tag in zip(rendered, order):
            out[k]["_gsr_stamps"] = resolver(slots[tag])
        return out

    # ------------------------------------------------------------------ validation (one iteration late)
    def _observe(self, entry):
        """feed what the replay measured to the planners of the eager path (capacities keep tracking the scene)"""
        if entry.ctx.counts is not None:
            planner, _ = self._planner_caps(entry.ctx.counts[2][2])
            host, _caps, shape = entry.ctx.counts
            if planner is not None:
                import numpy as np

                planner.observe(host.numpy().reshape(shape).astype(np.int64))

    def _check_inflight(self):
        """-> None when the replay in flight (if any) was fine; otherwise the result of repeating it eagerly"""
        infl, self._inflight = self._inflight, None
        if infl is None:
            return None
        entry, seq, ev, queue = infl
        slot = 2 * (seq % self.RING)
        ring = self._ring_np
        spins = 0
        while int(ring[slot + 1]) != seq:  # the stamp lands a few microseconds after the replay's last kernel
            spins += 1
            if spins > 2000:
                ev.synchronize()  # (also surfaces a faulted kernel)
                if int(ring[slot + 1]) != seq:
                    raise RuntimeError("graphed iteration: the replay finished without publishing its flag word")
        if int(ring[slot]) == 0:
            self._observe(entry)
            return None
        # a capacity did not hold in THAT replay: it and every later one were no-ops on the parameters (the flag is
        # sticky).  Let the device drain, clear the flag, drop the graphs (their capacities are stale) and repeat the
        # iterations eagerly, in order
        torch.cuda.current_stream().synchronize()
        self._flag.zero_()
        if self.opt.__dict__.get("_graph_owners") is not None:
            self.opt.graph_advance(-len(queue))
        self.entries.clear()
        self._seen.clear()
        out = None
        now = self._hyper_snapshot()
        for (cameras, strategies, tasks, hyper) in queue:
            self._hyper_restore(hyper)  # the learning rates / iteration counter of THAT iteration, not of the latest
            out = self.body(cameras, strategies, tasks)
            self.stats["redone"] += 1
        self._hyper_restore(now)
        return out

    def _hyper_snapshot(self):
        """what the host-side schedule has set for the iteration being launched: per-group learning rates and the
        process-wide iteration counter (a flagged replay is repeated eagerly LATER, when both have moved on)"""
        import utils.general_utils as utils

        return ([g["lr"] for g in self.opt.param_groups], utils.get_cur_iter())

    def _hyper_restore(self, hyper):
        import utils.general_utils as utils

        lrs, it = hyper
        for g, lr in zip(self.opt.param_groups, lrs):
            g["lr"] = lr
        if it is not None:
            utils.set_cur_iter(it)

    def reset(self):
        """drop every captured graph (after validating the iteration in flight): call when tensors a capture has baked in
        are replaced -- densification / pruning re-creates the parameters, Adam's moments and the statistics buffers;
        the key would notice the parameters' new address or shape, but an allocator that hands a freed address out again
        must never revive a stale graph.  The next iterations run eagerly (warm-up) and end with a new capture."""
        out = self._check_inflight()
        self.entries.clear()
        self._seen.clear()
        return out

    def validate(self):
        """wait for the iteration in flight and make sure it counted (repeating it eagerly if a capacity overflowed);
        call before reading results on the host.  -> the repeated iteration's result, or None"""
        return self._check_inflight()

    # ------------------------------------------------------------------ the step
    def __call__(self, cameras, strategies, tasks):
        self.last_stats = None
        if not self.enabled:
            self.stats["eager"] += 1
            return self.body(cameras, strategies, tasks)
        key = self._key(cameras, strategies)
        entry = self.entries.get(key)
        if entry is not None and not self._usable(entry, cameras, strategies):
            del self.entries[key]  # (the key has had its warm-up: the next eager iteration ends with a new capture)
            entry = None
        if entry is None:
            self._check_inflight()  # (a flagged replay is repeated here, before this iteration runs eagerly)
            self.stats["eager"] += 1
            out = self.body(cameras, strategies, tasks)
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n >= self.warmup:
                err = None
                try:
                    self._capture(self._key(cameras, strategies), cameras, strategies, tasks)
                except Exception as exc:  # noqa: BLE001
                    _dgr._CAPTURE[0] = None
                    err = f"{type(exc).__name__}: {exc}"
                # replaying is ONE decision of the whole group: a rank whose capture failed would run eagerly while its
                # peers replay, and their collectives would no longer pair up.  (Every rank reaches this point in the
                # same iteration: the key's sightings are functions of the partition, which all ranks share.)
                if self._group_says_no(err is not None):
                    self.entries.clear()
                    self.enabled = False
                    self.stats["disabled"] = err or "the capture failed on another rank"
            return out
        # replay: inputs + hyper-parameters (one host-to-device copy, one band copy per camera), then ONE launch
        seq = self._stage(entry, cameras, strategies)
        _dbg("replay", seq)
        entry.graph.replay()
        if _DEBUG:
            torch.cuda.synchronize()
            _dbg("replay done", seq, "flag", int(self._ring_np[2 * (seq % self.RING)]), "pairs",
                 [(int(h[0]), c) for h, c in entry.ctx.pairs])
        ev = torch.cuda.Event()
        ev.record()
        self.last_stats = self._replay_stats(entry, seq, ev, strategies)
        if self.opt.__dict__.get("_graph_owners") is not None:  # (a body without an optimizer step: nothing to count)
            self.opt.graph_advance(1)
        self.stats["replayed"] += 1
        prev = self._inflight
        if prev is None:
            self._inflight = (entry, seq, ev, [(cameras, strategies, tasks, self._hyper_snapshot())])
            return entry.out
        # look at the PREVIOUS replay now that this one keeps the device busy
        p_entry, p_seq, p_ev, p_queue = prev
        self._inflight = (p_entry, p_seq, p_ev, p_queue + [(cameras, strategies, tasks, self._hyper_snapshot())])
        redo = self._check_inflight()
        if redo is not None:
            self.last_stats = None  # (the iteration was repeated eagerly: the body's own stats carry its events)
            return redo
        self._inflight = (entry, seq, ev, [(cameras, strategies, tasks, self._hyper_snapshot())])
        return entry.out
