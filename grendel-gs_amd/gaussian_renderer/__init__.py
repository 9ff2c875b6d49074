"""Distributed render orchestration of the hot path (host side), MI355X build.

From-scratch mirror of the live (`*_final`) surface of the reference's gaussian_renderer/__init__.py
-- the names train_internal.py:5-10 and render.py:18-39 import -- on top of the HIP operator module
`diff_gaussian_rasterization` of this package:

  get_cuda_args_final                               gaussian_renderer/__init__.py:510-539
  all_to_all_communication_final                    gaussian_renderer/__init__.py:542-698
  distributed_preprocess3dgs_and_all2all_final      gaussian_renderer/__init__.py:878-1037
  render_final                                      gaussian_renderer/__init__.py:1217-1291
  render / preprocess3dgs_and_all2all (legacy)      gaussian_renderer/__init__.py:410-507
  gsplat_* twins                                    third-party backend, source absent -> raise

Same inputs, same returned dict keys / list shapes, same ordering of received Gaussians (source-rank
major, then local index).  What differs is HOW the one exchange step is done:
the reference issues W x B `nonzero()` host syncs, two all-to-alls (9 floats with autograd + 2 aux
floats) and per-destination cat/index kernels; here the per-band need masks of all cameras are stacked
into one [W, B, P] mask, ONE size all-gather + ONE host read-back sizes everything, and ONE
all-to-all-v (RCCL over xGMI: W-1 concurrent peer copies on the full mesh) carries an 11-float record
(means2D 2, rgb 3, conic_opacity 4, radius, depth); its autograd backward is the mirror all-to-all of
the 9 gradient columns followed by a scatter-add into the owners' per-camera buffers.  In the live
mode there is no gradient all-reduce at all.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

import utils.general_utils as utils
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

import diff_gaussian_rasterization as _dgr

import contextlib

_NULL = contextlib.nullcontext()


def _keep_grad_view(t):
    """What the reference's `means2D.retain_grad()` (gaussian_renderer/__init__.py:953) is there for -- densification
    reads `means2D.grad[:, :2]` (scene/gaussian_model.py:1046-1052) -- without the copy: retain_grad() CLONES the
    incoming gradient, which here is a column view of K10's [P,9] record (a strided 8-of-36-byte read: 9 us per
    iteration at 1 M Gaussians, 51 us at 6 M).  Nobody writes the record after K10 (K11 and the mirror exchange read
    it), so the view itself is kept as `.grad`.  Accumulates like retain_grad() if the tensor is used twice."""
    import weakref
    ref, first = weakref.ref(t), [True]

    def hook(g):
        x = ref()
        if x is not None:  # (reading .grad of a non-leaf that has none warns: the first call is remembered instead)
            x.grad = g if first[0] else x.grad + g
            first[0] = False

    t.register_hook(hook)

N_DIFF = 9  # means2D (2) + rgb (3) + conic_opacity (4): columns that carry gradients
N_AUX = 2   # radius (as float) + depth: no gradient


def get_cuda_args_final(strategy, mode="train"):
    args = utils.get_args()
    iteration = utils.get_cur_iter()
    if mode == "train":
        # a batch covers iterations [it, it + bsz): log on the one that hits the interval
        for x in range(args.bsz):
            if (iteration + x) % args.log_interval == 1:
                iteration += x
                break
    elif mode == "test":
        iteration = -1
    else:
        raise ValueError("mode should be train or test.")
    return {
        "mode": mode,
        "world_size": str(utils.WORLD_SIZE),
        "global_rank": str(utils.GLOBAL_RANK),
        "local_rank": str(utils.LOCAL_RANK),
        "mp_world_size": str(strategy.world_size),
        "mp_rank": str(strategy.rank),
        "log_folder": args.log_folder,
        "log_interval": str(args.log_interval),
        "iteration": str(iteration),
        "zhx_debug": str(args.zhx_debug),
        "zhx_time": str(args.zhx_time),
        "avoid_pixel_all2all": False,
        "stats_collector": {},
    }


# ------------------------------------------------------------------------------ the exchange
class _SparseExchange(torch.autograd.Function):
    """rows `send_idx` of `diff` (+ the same rows of `aux`) -> all-to-all-v -> received records.
    backward: mirror all-to-all of the gradient columns, scatter-add into the senders' rows (a
    Gaussian needed by two bands gets two contributions)."""

    @staticmethod
    def forward(ctx, diff, aux, send_idx, send_splits, recv_splits, group):
        msg = torch.cat([diff.index_select(0, send_idx), aux.index_select(0, send_idx)], dim=1).contiguous()
        recv = torch.empty((sum(recv_splits), msg.shape[1]), dtype=msg.dtype, device=msg.device)
        dist.all_to_all_single(recv, msg, output_split_sizes=recv_splits, input_split_sizes=send_splits, group=group)
        ctx.group, ctx.send_splits, ctx.recv_splits = group, send_splits, recv_splits
        ctx.n_rows = diff.shape[0]
        ctx.save_for_backward(send_idx)
        out_diff = recv[:, :N_DIFF].contiguous()
        out_aux = recv[:, N_DIFF:].contiguous()
        ctx.mark_non_differentiable(out_aux)
        return out_diff, out_aux

    @staticmethod
    def backward(ctx, g_diff, _g_aux):
        (send_idx,) = ctx.saved_tensors
        g_diff = g_diff.contiguous()
        back = torch.empty((sum(ctx.send_splits), N_DIFF), dtype=g_diff.dtype, device=g_diff.device)
        dist.all_to_all_single(back, g_diff, output_split_sizes=ctx.send_splits, input_split_sizes=ctx.recv_splits,
                               group=ctx.group)
        grad = torch.zeros((ctx.n_rows, N_DIFF), dtype=g_diff.dtype, device=g_diff.device)
        grad.index_add_(0, send_idx, back)
        return grad, None, None, None, None, None


def all_to_all_communication_final(batched_rasterizers, batched_screenspace_params, batched_cuda_args,
                                   batched_strategies):
    """-> (means2D, rgb, conic_opacity, radii, depths) lists per camera of what THIS rank must render
    (rows ordered by source rank, then by the source's local index), + gpui_to_gpuj_imgk_size[W][W][B]"""
    group = utils.DEFAULT_GROUP
    W, me = group.size(), group.rank()
    B = len(batched_rasterizers)
    P = batched_screenspace_params[0][0].shape[0]
    dev = batched_screenspace_params[0][0].device

    # need[g, k, i]: global rank g renders a band of camera k that Gaussian i touches
    need = torch.zeros((W, B, P), dtype=torch.bool, device=dev)
    for k, strategy in enumerate(batched_strategies):
        means2D, _, _, radii, _ = batched_screenspace_params[k]
        rs = batched_rasterizers[k].raster_settings
        band_mask = strategy.get_local2j_ids_bool(means2D, radii, rs.image_height, rs.image_width,
                                                  batched_cuda_args[k])
        for j, g in enumerate(strategy.gpu_ids):
            need[g, k] = band_mask[:, j]

    counts = need.sum(dim=2, dtype=torch.int32)  # [W(dst), B]
    all_counts = torch.empty((W * W, B), dtype=torch.int32, device=dev)  # dim-0 concatenation of the [W, B] inputs
    dist.all_gather_into_tensor(all_counts, counts.contiguous(), group=group)
    sizes = all_counts.view(W, W, B).cpu().tolist()  # the one host read-back of the exchange; sizes[i][j][k]
    send_splits = [sum(sizes[me][j]) for j in range(W)]
    recv_splits = [sum(sizes[i][me]) for i in range(W)]

    # rows of the [B*P, .] state matrix to send, already in (dst, camera, local index) order
    flat = torch.nonzero_static(need.view(-1), size=sum(send_splits)).squeeze(1)
    send_idx = flat % (B * P)  # (k, i) -> k * P + i

    diff = torch.cat([torch.cat([p[0], p[1], p[2]], dim=1) for p in batched_screenspace_params], dim=0)
    aux = torch.cat([torch.stack([p[3].to(diff.dtype), p[4]], dim=1) for p in batched_screenspace_params], dim=0)
    r_diff, r_aux = _SparseExchange.apply(diff, aux.detach(), send_idx, send_splits, recv_splits, group)

    # received rows are (src, camera, ...)-major; regroup per camera keeping the source-rank order
    seg = [sizes[i][me][k] for i in range(W) for k in range(B)]
    d_parts = torch.split(r_diff, seg, dim=0)
    a_parts = torch.split(r_aux, seg, dim=0)
    out = ([], [], [], [], [])
    for k in range(B):
        d = torch.cat([d_parts[i * B + k] for i in range(W)], dim=0)
        a = torch.cat([a_parts[i * B + k] for i in range(W)], dim=0)
        out[0].append(d[:, 0:2])
        out[1].append(d[:, 2:5])
        out[2].append(d[:, 5:9])
        out[3].append(a[:, 0].int())
        out[4].append(a[:, 1])
    return out[0], out[1], out[2], out[3], out[4], sizes


# ---- the fused exchange of the camera-batched state -------------------------------------------------------------
# switches (parity tests flip them):
#   overlap   : pipeline the per-camera exchanges on a side HIP stream
#   speculate : pack into capacity slabs chosen from earlier iterations, so that this iteration's counts never have to
#               reach the host before the all-to-all is launched (no `.cpu()` between K1 and the render)
#   group     : ONE exchange for the whole batch when every rank renders (a band of) at most one of its cameras
_EXCHANGE_OPTIONS = {"overlap": True, "speculate": True, "forced": False,
                     "group": os.environ.get("GSR_EXCHANGE_GROUP", "1") != "0",  # (env: A/B measurements only)
                     "camera_streams": int(os.environ.get("GSR_CAMERA_STREAMS", "1") or 1),
                     "single_thread_backward": os.environ.get("GSR_AUTOGRAD_MULTITHREAD", "0") != "1"}
_SIDE_STREAMS = {}
_BANDS_CACHE = {}
_CAMERA_BLOCKS = {}  # ids of a batch's packed camera records -> (their [B,40] stack, the records, their versions)
_PLANNERS = {}
exchange_stats = {"speculative": 0, "sized": 0, "redone": 0}  # how the exchanges of this process were laid out


def set_exchange_overlap(enabled):
    """camera k's all-to-all (pack, RCCL, unpack -- and its mirror in the backward) on a side HIP stream, so that it
    runs beside camera k-1's K3-K8 in the forward and beside camera k+1's K10 in the backward (north_star); off: one
    exchange for the whole batch on the current stream"""
    _EXCHANGE_OPTIONS["overlap"] = bool(enabled)


def set_camera_streams(n):
    """n = 2: the cameras a rank renders in one batch (bsz > 1) alternate between two side HIP streams, so that camera
    k + 1's K3-K7 (latency chains) run beside camera k's K8 / loss kernels (VALU-bound) in the forward and the K10s of
    two cameras beside each other in the backward (autograd runs a node's backward on its forward's stream); images
    and gradients are those of the one-stream order (the kernels and their inputs are the same).  1 (default): one
    stream, the reference's order (gaussian_renderer/__init__.py:1217-1291 renders its cameras one after the other)."""
    _EXCHANGE_OPTIONS["camera_streams"] = 2 if int(n) >= 2 else 1


def set_single_thread_backward(enabled):
    """True (default; GSR_AUTOGRAD_MULTITHREAD=1 in the environment: False): the training thread's backward runs the
    autograd nodes of this package ON THAT THREAD.  PyTorch's engine hands a CUDA graph's nodes to a per-device worker
    thread; every node of the hot path is a Python function (one or two C-ABI launches each), so every node is a
    hand-over of the GIL -- measured on one rank of 8 (configs[2]'s shape): the eager iteration 1.14-1.34 -> 0.98 ms, and
    the step time no longer moves with the box's host.  One process drives one GPU here (utils/general_utils.py:194-235 of
    the reference), which is the case the worker threads do nothing for.  The setting is PyTorch's own
    torch.autograd.set_multithreading_enabled -- thread-local -- applied at the top of every
    distributed_preprocess3dgs_and_all2all_final(mode="train") on the calling thread."""
    _EXCHANGE_OPTIONS["single_thread_backward"] = bool(enabled)


def set_exchange_grouping(enabled):
    """True (default): when every rank renders a band of AT MOST ONE camera of the batch (bsz <= world size, the
    reference's usual batched mode: train_internal.py with bsz 4 or 8 on 4-8 GPUs), the capacity-slab exchange of the
    whole batch is ONE pack, ONE all-to-all-v and ONE unpack each way instead of one per camera.  Nothing needs
    regrouping then: the rows a rank receives for the cameras it renders no part of are the all-zero padding records of
    their slabs (radius 0: K3 culls them), so the whole message IS its camera's input, in the reference's (source,
    index) order.  Per-camera exchanges cannot overlap anything on such a rank (its one render needs its one camera's
    rows, its K10 produces that camera's gradients), they only multiply the launches of a host-bound step by bsz.
    False, or a rank that renders two cameras: one exchange per camera, pipelined on the side stream."""
    _EXCHANGE_OPTIONS["group"] = bool(enabled)


def set_exchange_speculation(enabled):
    """True (default): capacity slabs, no host read-back of this iteration's counts before the all-to-all (they are
    verified after the first render has polled its pair count, and the exchange is repeated with exact sizes if a slab
    overflowed).  False: the reference's order -- all-gather the counts, read them back, size everything exactly
    (gaussian_renderer/__init__.py:572-585)."""
    _EXCHANGE_OPTIONS["speculate"] = bool(enabled)


def set_exchange_forced(enabled):
    """run the exchange in a ONE-rank group too (every visible row is "sent" to the rank itself through the real
    collectives): the single-GPU tests and tools/overlap_trace.py exercise pack / RCCL / unpack / verification / the
    mirror backward this way.  Off (default): world size 1 hands K1's outputs straight to the renderer, as the
    reference does (gaussian_renderer/__init__.py:969-1000)."""
    _EXCHANGE_OPTIONS["forced"] = bool(enabled)


def _side_stream(dev, purpose="exchange"):
    key = (dev.type, dev.index, purpose)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _bands_tensor(batched_strategies, W, dev):
    """int32 [B, W, 2] tile rows [lo, hi) of camera k rendered by global rank g, on the device; cached per partition
    (static strategies -- bsz >= W, frozen heuristics -- hit the cache every step: no host-to-device copy)"""
    dyn = getattr(batched_strategies[0], "_gsr_dyn_all_bands", None) if batched_strategies else None
    if dyn is not None:  # a band-agnostic hipGraph capture (graphed_step.py): the table is refreshed in front of a replay
        return dyn
    key = (dev.index, W) + tuple((tuple(s.gpu_ids), tuple(s.division_pos)) for s in batched_strategies)
    t = _BANDS_CACHE.get(key)
    if t is None:
        if len(_BANDS_CACHE) > 512:
            _BANDS_CACHE.clear()
        bands = [[(0, 0)] * W for _ in batched_strategies]
        for k, strategy in enumerate(batched_strategies):
            for j, g in enumerate(strategy.gpu_ids):
                bands[k][g] = (strategy.division_pos[j], strategy.division_pos[j + 1])
        t = _BANDS_CACHE[key] = torch.tensor(bands, dtype=torch.int32).to(dev)
    return t


def _camera_major_base(per_camera):
    """the dense [B,P,.] buffer whose slices the per-camera tensors are (the batched K1 allocates them that way), or a
    stacked copy"""
    t0 = per_camera[0]
    base = t0._base
    if (base is not None and base.is_contiguous() and base.shape[0] == len(per_camera)
            and tuple(base.shape[1:]) == tuple(t0.shape)
            and all(t._base is base and t.data_ptr() == base.data_ptr() + k * t0.numel() * t0.element_size()
                    for k, t in enumerate(per_camera))):
        return base.detach()
    return torch.stack([t.detach() for t in per_camera]).contiguous()


class _SlabPlanner:
    """Capacities of the read-back-free exchange: caps[i][j][k] rows are reserved for what rank i sends rank j of the
    batch's k-th camera.  A pure function of the all-gathered count matrices of EARLIER iterations -- every rank holds
    the same matrices, so every rank derives the same capacities and takes the same overflow decision without any
    extra collective.  capacity = 1.25 x (largest count of the last 64 iterations) + 256, rounded up to 256 rows: the
    exchange is latency-bound (~12 us per peer at link rate), padding rows cost 44 B each."""

    WINDOW = 64

    def __init__(self, W, B):
        self.W, self.B = W, B
        self._hist, self._n = None, 0   # ring of the last WINDOW count matrices (numpy: these are W x W x B numbers)
        self.caps = None        # int64 numpy [W, W, B]
        self.caps_list = None   # the same as nested python lists
        self.pending = None     # (event | None, host matrix, caps it was packed with) of an unverified iteration
        self._ring, self._slot = [], 0
        self.redo_streak = 0    # consecutive speculative exchanges that had to be repeated ...
        self.backoff = 0        # ... after REDO_LIMIT of them: this many iterations with the exact layout

    REDO_LIMIT, BACKOFF = 3, 32

    def note(self, redone):
        """a scene / partition where a rendered band keeps receiving < 10 rows (an empty sky band, early training) or the
        counts keep outgrowing their slabs would pay two exchanges and two renders every iteration: after REDO_LIMIT
        repeats in a row the next BACKOFF iterations use the exact layout (one read-back, the reference's schedule).  The
        same decision on every rank: `redone` is a function of the all-gathered counts."""
        self.redo_streak = self.redo_streak + 1 if redone else 0
        if self.redo_streak >= self.REDO_LIMIT:
            self.redo_streak, self.backoff = 0, self.BACKOFF

    def observe(self, m):
        """m: int64 numpy array [W, W, B]"""
        if self._hist is None:
            self._hist = np.zeros((self.WINDOW,) + m.shape, dtype=np.int64)
            self._n = 0
        self._hist[self._n % self.WINDOW] = m
        self._n += 1
        peak = self._hist[:min(self._n, self.WINDOW)].max(axis=0)
        want = (peak * 5 // 4 + 256 + 255) // 256 * 256
        if self.caps is None or (want > self.caps).any() or (want * 2 < self.caps).any():
            self.caps = want
            self.caps_list = want.tolist()

    def stage(self, all_counts):
        """asynchronous copy of this iteration's all-gathered counts to the host; nothing waits for it here"""
        if all_counts.is_cuda:
            if len(self._ring) < 4:
                self._ring.append(torch.empty(all_counts.shape, dtype=all_counts.dtype).pin_memory())
            host = self._ring[self._slot % len(self._ring)] if len(self._ring) == 4 else self._ring[-1]
            self._slot += 1
            host.copy_(all_counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(_dgr.current_stream(all_counts.device))
        else:
            host, ev = all_counts.clone(), None
        self.pending = (ev, host, self.caps)
        return self.pending

    def resolve(self, pending=None):
        """-> (count matrix int64 numpy [W, W, B] of `pending` (default: the latest staged iteration), True when every
        count fitted its slab).  A package keeps ITS pending tuple, so a second package created before the first one is
        verified cannot be mistaken for it."""
        if pending is None:
            pending = self.pending
        if pending is self.pending:
            self.pending = None
        ev, host, caps = pending
        if ev is not None:
            ev.synchronize()  # the copy was queued before the all-to-all: long complete when a render has polled D
        m = host.numpy().reshape(self.W, self.W, self.B).astype(np.int64)  # (a copy: the pinned buffer is reused)
        if (m < 0).any():
            raise RuntimeError("exchange: negative row counts read back (a missing stream dependency in the collective?)")
        self.observe(m)
        return m, bool((m <= caps).all())


def _planner(group, W, B):
    key = (id(group), W, B)
    p = _PLANNERS.get(key)
    if p is None:
        p = _PLANNERS[key] = _SlabPlanner(W, B)
    return p


class _ExchangeGroup(torch.autograd.Function):
    """the exchange of the cameras [k0, k0 + nb) of a batch: pack (gsr_exchange_pack / gsr_exchange_pack_slab: 11-float
    records in (destination, camera, local index) order, straight into the send buffer), ONE all-to-all-v, unpack.  Runs
    on the stream that is current when it is called -- autograd runs the backward (gradient rows back through the mirror
    all-to-all-v, gsr_scatter_add_rows into the owners' rows) on the same stream.  Differentiable inputs: the per-camera
    means2D / rgb / conic_opacity views (their gradients come back as column views of one [nb*P, 9] record that the
    batched K11 reads through its row stride)."""

    @staticmethod
    def forward(ctx, meta, bases, token, *views):
        (k0, nb, P, width, height, layout, send_splits, recv_splits, perm, inv_perm, group, chunkcnt, counts, cnt_B,
         bands, consumer_stream, holder) = meta
        # (unused outputs -- radii, depths, the token -- arrive in backward as None, not as zero tensors the engine would
        # fill first: two n_recv-sized fill launches per exchange otherwise)
        ctx.set_materialize_grads(False)
        m2_all, rgb_all, co_all, radii_all, depths_all = bases
        dev = radii_all.device
        n_send, n_recv = sum(send_splits), sum(recv_splits)
        cur = _dgr.current_stream(dev) if (dev.type == "cuda" and consumer_stream is not None) else None
        if layout[0] == "slab":  # capacities per (destination, camera); this iteration's counts stay on the device
            msg, send_idx = _dgr.exchange_pack_slab(m2_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt,
                                                    counts, layout[1], k0, nb, width, height, count_cameras=cnt_B,
                                                    count_first=0)
        else:                    # exact segment offsets from counts the host has read
            msg, send_idx = _dgr.exchange_pack(m2_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt,
                                               layout[1], n_send, k0, nb, width, height, count_cameras=cnt_B,
                                               count_first=0)
        recv = torch.empty((n_recv, N_DIFF + N_AUX), dtype=msg.dtype, device=dev)
        dist.all_to_all_single(recv, msg, output_split_sizes=recv_splits, input_split_sizes=send_splits, group=group)
        if perm is None:  # one exchange per camera (always, with capacity slabs): the message is camera-major already
            outs = list(_dgr.exchange_unpack(recv))
        else:             # several cameras in one exact exchange: (source, camera)-major -> camera-major
            outs = [torch.empty((n_recv, w), dtype=msg.dtype, device=dev) for w in (2, 3, 4, 1, 1)]
            _dgr.gather_rows(perm, n_recv, [recv[:, 0:2], recv[:, 2:5], recv[:, 5:9], recv[:, 9:10], recv[:, 10:11]],
                             outs)
        if consumer_stream is not None and consumer_stream != cur:
            for t in outs:
                t.record_stream(consumer_stream)  # allocated on the side stream, consumed by the renderer's stream
            for t in bases:
                t.record_stream(cur)
        ctx.meta = (k0, nb, P, cnt_B, send_splits, recv_splits, group, consumer_stream, holder)
        ctx.fdt = msg.dtype  # the gradient message's type on EVERY rank, also one whose three gradients all are None
        ctx.save_for_backward(send_idx, inv_perm if inv_perm is not None else send_idx[:0])
        ctx.has_perm = inv_perm is not None
        r_radii = outs[3] if outs[3].dtype == torch.int32 else (
            outs[3].view(torch.int32) if outs[3].dtype == torch.float32 else outs[3].to(torch.int32))
        r_radii = r_radii.reshape(n_recv)
        r_depths = outs[4].reshape(n_recv)
        ctx.mark_non_differentiable(r_radii, r_depths)
        # the token chains the per-camera exchanges of a batch: camera k's node consumes camera k-1's token, the last
        # one is handed to a render op, so every rank's backward runs the mirror collectives of ALL cameras (also of
        # those it renders no part of) and in the same order, camera B-1 first.  Its VALUE is read by the < 10-Gaussian
        # stand-in of render_final (0 * token): it must be a defined zero, not an uninitialised word
        token_out = None
        if token is not None:
            token_out = torch.zeros((1,), dtype=outs[0].dtype, device=dev)
            if consumer_stream is not None and consumer_stream != cur:
                token_out.record_stream(consumer_stream)
        return outs[0], outs[1], outs[2], r_radii, r_depths, token_out

    @staticmethod
    def backward(ctx, g_m2, g_rgb, g_co, _gr, _gd, _gtoken):
        send_idx, inv_perm = ctx.saved_tensors
        k0, nb, P, B, send_splits, recv_splits, group, consumer_stream, holder = ctx.meta
        n_recv, n_send = sum(recv_splits), sum(send_splits)
        dev = send_idx.device
        cur = _dgr.current_stream(dev) if (dev.type == "cuda" and consumer_stream is not None) else None
        fdt = ctx.fdt  # fp32 (fp64 in the CPU tests)
        g_recv = None
        if not ctx.has_perm and all(g is not None and g.dtype == torch.float32 for g in (g_m2, g_rgb, g_co)):
            base = g_m2._base
            if (base is not None and base.is_contiguous() and tuple(base.shape) == (n_recv, N_DIFF)
                    and g_rgb._base is base and g_co._base is base and g_m2.data_ptr() == base.data_ptr()
                    and g_rgb.data_ptr() == base.data_ptr() + 8 and g_co.data_ptr() == base.data_ptr() + 20):
                g_recv = base  # K10's gradient record has the message's column order: it IS the message
        if g_recv is None:
            gs = [g if g is not None else torch.zeros((n_recv, w), dtype=fdt, device=dev)
                  for g, w in ((g_m2, 2), (g_rgb, 3), (g_co, 4))]
            gs = [g if (g.dtype == fdt and (g.shape[0] <= 1 or g.stride(1) == 1)) else g.to(fdt).contiguous()
                  for g in gs]
            g_recv = torch.empty((n_recv, N_DIFF), dtype=fdt, device=dev)
            _dgr.gather_rows(inv_perm if ctx.has_perm else None, n_recv, gs,
                             [g_recv[:, 0:2], g_recv[:, 2:5], g_recv[:, 5:9]])
        if consumer_stream is not None and consumer_stream != cur:
            g_recv.record_stream(cur)  # produced by K10 on the renderer's stream, read here on the side stream
        back = torch.empty((n_send, N_DIFF), dtype=fdt, device=dev)
        dist.all_to_all_single(back, g_recv, output_split_sizes=send_splits, input_split_sizes=recv_splits, group=group)
        # ONE [B*P, 9] record for the whole batch (the exchanges of its cameras add into their own row blocks); the
        # gradients that go back are its column views, which the batched K11 reads through the row stride
        if holder.get("rec") is None:
            holder["rec"] = _dgr.zeros_async((B * P, N_DIFF), fdt, dev)
            if consumer_stream is not None and consumer_stream != cur:
                holder["rec"].record_stream(consumer_stream)
        rec = holder["rec"][k0 * P:(k0 + nb) * P]
        _dgr.scatter_add_rows(send_idx, back, nb * P, dst=rec)  # slab padding rows carry index -1: skipped
        grads = []
        for a, b in ((0, 2), (2, 5), (5, 9)):
            grads += [rec[k * P:(k + 1) * P, a:b] for k in range(nb)]
        return (None, None, None) + tuple(grads)


class _LazyPerCamera:
    """a list whose k-th entry is fn(source[k]), evaluated on first access (one kernel launch per camera and iteration
    that only densification steps need)"""

    def __init__(self, source, fn):
        self._src, self._fn, self._v = source, fn, [None] * len(source)

    def __len__(self):
        return len(self._src)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self._src)))]
        if self._v[k] is None:
            self._v[k] = self._fn(self._src[k])
        return self._v[k]

    def __iter__(self):
        return (self[k] for k in range(len(self._src)))


class _LazySizes:
    """gpui_to_gpuj_imgk_size of a speculative exchange: the [W][W][B] python ints exist once the asynchronous copy of
    the counts has been looked at (render_final does); list-like for the code that reads them afterwards"""

    def __init__(self):
        self._v = None
        self._resolver = None

    def _get(self):
        if self._v is None:
            self._resolver()
        return self._v

    def __getitem__(self, i):
        return self._get()[i]

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())

    def __eq__(self, other):
        return self._get() == (other._get() if isinstance(other, _LazySizes) else other)

    def __repr__(self):
        return repr(self._get())


def _batched_exchange_final(m2_views, rgb_views, co_views, radii_views, depths_views, rasterizers,
                            batched_strategies, speculate=None, pipeline=None, _known=None):
    """all_to_all_communication_final for the camera-batched state: same return structure (+ the per-camera events a
    consumer on another stream must wait for, None without overlap; + the pending verification of a speculative
    layout, None for an exact one).  Host cost per batch: one count launch, one size all-gather, then per exchange one
    pack launch, one all-to-all-v and one unpack launch -- and, only when the layout is not speculative, the one
    read-back of the counts the reference has too (gaussian_renderer/__init__.py:572-585)."""
    group = utils.DEFAULT_GROUP
    if group.size() == 1 and not isinstance(group, dist.ProcessGroup):
        group = dist.group.WORLD  # set_exchange_forced in a process whose DEFAULT_GROUP is the one-rank stand-in
    W, me = group.size(), group.rank()
    B = len(radii_views)
    bases = [_camera_major_base(v) for v in (m2_views, rgb_views, co_views, radii_views, depths_views)]
    radii_all = bases[3]
    P = radii_all.shape[1]
    dev = radii_all.device
    rs = rasterizers[0].raster_settings
    width, height = int(rs.image_width), int(rs.image_height)
    bands = _bands_tensor(batched_strategies, W, dev)
    planner = _planner(group, W, B)
    if planner.pending is not None:
        # a speculative exchange whose consumer never went through render_final: look at its counts now
        _, fitted = planner.resolve()
        if not fitted:
            raise RuntimeError("the previous iteration's speculative exchange overflowed a slab and was consumed "
                               "without render_final()'s verification; call gaussian_renderer.render_final or "
                               "set_exchange_speculation(False)")
    speculate = _EXCHANGE_OPTIONS["speculate"] if speculate is None else speculate
    gather_work = None
    cap_ctx = _dgr.capturing()
    if cap_ctx is not None:
        # the iteration is being captured in a hipGraph (graphed_step.py): capacity slabs are the only layout that needs
        # nothing from the host; the verification happens ON THE DEVICE -- gsr_exchange_check raises the capture's flag
        # word when a count exceeds its slab or a rendered band receives fewer than 10 rows, which makes the optimizer
        # launch of the same replay a no-op -- and the host looks at the counts (copied to pinned memory by the graph)
        # after the replay
        if planner.caps is None or _known is not None or cap_ctx.slab_caps_dev is None:
            raise RuntimeError("graph capture: the exchange needs slab capacities from earlier eager iterations")
        speculate = True
        chunkcnt, counts = _dgr.exchange_count(bases[0], radii_all, bands, 0, B, width, height)
        all_counts = torch.empty((W * W, B), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(all_counts, counts, group=group)
        rendered = 0
        for k, st in enumerate(batched_strategies):
            if me in st.gpu_ids:
                rendered |= 1 << k
        host = cap_ctx.take_pinned(W * W * B)  # the check kernel leaves the counts there (pinned, device-accessible)
        _dgr.exchange_check(all_counts, cap_ctx.slab_caps_dev, W, B, me, rendered, 10, cap_ctx.flag, host_copy=host)
        cap_ctx.counts = (host, planner.caps.copy(), (W, W, B))
        sizes = _LazySizes()
        sizes._resolver = lambda h=host, z=sizes: setattr(z, "_v", h.view(W, W, B).tolist())
    elif _known is not None:   # the repeat of an overflowed speculative exchange: counts already on the host
        chunkcnt, counts, sizes = _known
        speculate = False
    else:
        chunkcnt, counts = _dgr.exchange_count(bases[0], radii_all, bands, 0, B, width, height)
        all_counts = torch.empty((W * W, B), dtype=torch.int32, device=dev)
        if planner.backoff > 0:  # repeated redos: exact layout for a while (see _SlabPlanner.note)
            planner.backoff -= 1
            speculate = False
        speculate = speculate and planner.caps is not None
        if speculate and dev.type == "cuda":
            # nothing the pack / all-to-all / unpack below need depends on the gathered matrix (the pack works from the
            # LOCAL counts): the collective runs on its own stream beside them, and this stream only joins it -- and
            # queues the copy to the host -- AFTER the unpack has been enqueued (see the end of this function)
            gather_work = dist.all_gather_into_tensor(all_counts, counts, group=group, async_op=True)
            sizes = _LazySizes()
        elif speculate:
            dist.all_gather_into_tensor(all_counts, counts, group=group)
            planner.stage(all_counts)
            sizes = _LazySizes()
        else:
            dist.all_gather_into_tensor(all_counts, counts, group=group)
            sizes = all_counts.view(W, W, B).cpu().tolist()  # the one host read-back of an exact exchange; sizes[i][j][k]
            planner.observe(np.asarray(sizes, dtype=np.int64).reshape(W, W, B))
    exchange_stats["speculative" if speculate else "sized"] += 1
    layout_sizes = planner.caps_list if speculate else sizes  # rows per (source, destination, camera) of the buffers

    pipeline = _EXCHANGE_OPTIONS["overlap"] if pipeline is None else pipeline
    # one exchange for the whole batch (set_exchange_grouping): the same decision on every rank -- it is a function of
    # the strategies, which every rank derives from the same heuristics
    renders = [0] * W
    for st in batched_strategies:
        for g in st.gpu_ids:
            renders[g % W] += 1
    grouped = bool(speculate and B > 1 and _EXCHANGE_OPTIONS["group"] and max(renders) <= 1)
    pipelined = (pipeline or speculate) and B > 1 and not grouped   # one exchange per camera ...
    # ... on the side stream -- not inside a hipGraph capture: the process group's watchdog thread polls the events of
    # collectives issued from a stream that has not joined the capture yet, which HIP forbids (measured: the capture
    # aborts with hipErrorCapturedEvent); a captured iteration keeps its exchanges on the capturing stream
    overlap = pipelined and pipeline and dev.type == "cuda" and cap_ctx is None
    groups = [(k, 1) for k in range(B)] if pipelined else [(0, B)]
    cur = _dgr.current_stream(dev) if dev.type == "cuda" else None
    side = _side_stream(dev) if overlap else None
    if overlap:
        side.wait_stream(cur)  # K1's outputs and the counts are ready
        if not _EXCHANGE_OPTIONS.get("_persist_off"):
            # A collective of camera k+1 runs on the side stream WHILE camera k is binned and rendered.  The persistent
            # prepare kernel (csrc/binning_persist.h) needs its whole grid resident: behind an all-to-all that waits for a
            # late peer it would sit at its first barrier until its 50 ms time-out sends the view to the look-back
            # pipeline anyway.  While exchanges overlap, the binning therefore takes the look-back pipeline (many small
            # launches that run beside the collective) -- unless the caller chose a mode (set_bin_persistent) or the
            # environment names one (GSR_BIN_PERSIST); the switch is undone by the first batch without overlap (below).
            if _dgr.bin_persistent_user_choice() is None and not os.environ.get("GSR_BIN_PERSIST"):
                _dgr.set_bin_persistent("off", _internal=True)
                _EXCHANGE_OPTIONS["_persist_off"] = True
    elif _EXCHANGE_OPTIONS.get("_persist_off") and cap_ctx is None:
        if _dgr.bin_persistent_user_choice() is None:
            _dgr.set_bin_persistent("env", _internal=True)
        _EXCHANGE_OPTIONS["_persist_off"] = False
    out = ([None] * B, [None] * B, [None] * B, [None] * B, [None] * B)
    events = [None] * B
    holder = {"rec": None}  # the backward's shared [B*P, 9] gradient record
    token = torch.zeros((1,), dtype=bases[0].dtype, device=dev, requires_grad=True) if pipelined else None
    for (k0, nb) in groups:
        cams = range(k0, k0 + nb)
        send_splits = [sum(layout_sizes[me][j][k] for k in cams) for j in range(W)]
        recv_splits = [sum(layout_sizes[i][me][k] for k in cams) for i in range(W)]
        per_cam = [sum(layout_sizes[i][me][k] for i in range(W)) for k in cams]
        perm = inv_perm = None
        if speculate:
            layout = ("slab", [layout_sizes[me][g][k] for g in range(W) for k in cams])
        else:
            seg_off, o = [0] * (W * nb), 0
            for g in range(W):
                for kk, k in enumerate(cams):
                    seg_off[g * nb + kk] = o
                    o += sizes[me][g][k]
            layout = ("sized", seg_off)
            # received rows are (source, camera)-major; the renderer wants camera-major with the source order kept
            if sum(1 for n in per_cam if n) > 1:
                # segments (source i, camera k) arrive source-major; list them camera-major.  The row permutation is
                # expanded ON THE DEVICE from the W * nb segment descriptors (a few hundred bytes of host data)
                off, o = {}, 0
                for i in range(W):
                    for k in cams:
                        off[(i, k)] = o
                        o += sizes[i][me][k]
                segs = [(off[(i, k)], sizes[i][me][k]) for k in cams for i in range(W) if sizes[i][me][k] > 0]
                total = sum(n for _, n in segs)
                lens = torch.tensor([n for _, n in segs], dtype=torch.int64)
                dst0 = torch.cumsum(lens, 0) - lens
                shift = (torch.tensor([b for b, _ in segs], dtype=torch.int64) - dst0).to(dev)
                order = (torch.repeat_interleave(shift, lens.to(dev), output_size=total)
                         + torch.arange(total, dtype=torch.int64, device=dev))
                inv = torch.empty(total, dtype=torch.int32, device=dev)
                inv[order] = torch.arange(total, dtype=torch.int32, device=dev)
                perm, inv_perm = order.to(torch.int32), inv
        meta = (k0, nb, P, width, height, layout, send_splits, recv_splits, perm, inv_perm, group, chunkcnt, counts, B,
                bands, cur if overlap else None, holder)
        views = [m2_views[k] for k in cams] + [rgb_views[k] for k in cams] + [co_views[k] for k in cams]
        if overlap:
            with torch.cuda.stream(side):
                r = _ExchangeGroup.apply(meta, bases, token, *views)
                ev = torch.cuda.Event()
                ev.record(side)
        else:
            r, ev = _ExchangeGroup.apply(meta, bases, token, *views), None
        token = r[5]
        if grouped:  # the whole message is the input of the one camera this rank renders (the rest: zero records)
            for k in cams:
                local = me in batched_strategies[k].gpu_ids
                for c in range(5):
                    out[c][k] = r[c] if local else r[c][:0]
            continue
        start = 0
        for kk, k in enumerate(cams):
            n = per_cam[kk]
            whole = n == r[3].shape[0]
            for c in range(5):
                out[c][k] = r[c] if whole else r[c][start:start + n]
            events[k] = ev
            start += n
    if gather_work is not None:
        gather_work.wait()          # stream-level join with the (long finished) size all-gather; the host does not wait
        planner.stage(all_counts)   # asynchronous copy to pinned memory + event, behind the unpack in stream order
    pending = (planner, chunkcnt, counts, sizes, planner.pending) if (speculate and cap_ctx is None) else None
    return out[0], out[1], out[2], out[3], out[4], sizes, (events, token), pending


def distributed_preprocess3dgs_and_all2all_final(batched_viewpoint_cameras, pc, pipe, bg_color, scaling_modifier=1.0,
                                                 batched_strategies=None, mode="train", _legacy=False):
    """every rank projects ITS shard of Gaussians for EVERY camera of the batch (K1), then the sparse
    exchange hands each rank the Gaussians touching the bands it renders.  (`_legacy`: set by the legacy single-camera
    wrapper below, whose render() has no verification / stream hand-over: exact sizes, one exchange, current stream.)"""
    timers = utils.get_timers()
    args = utils.get_args()
    assert utils.DEFAULT_GROUP.size() == 1 or (args.gaussians_distribution and args.image_distribution), \
        "Ensure distributed training given multiple GPU. "
    if mode == "train" and _EXCHANGE_OPTIONS["single_thread_backward"] and torch.autograd.is_multithreading_enabled():
        torch.autograd.set_multithreading_enabled(False)  # (thread-local; see set_single_thread_backward)

    # the operator records its render timings (HIP events, resolved by finish_strategy_final) only when the load
    # balancer or a saved strategy history will read them; asked for per call (cuda_args), not through the module-wide
    # timing mode a user may have set
    from gaussian_renderer.workload_division import timings_have_consumer
    timing = "deferred" if (mode != "train" or timings_have_consumer()) else "off"
    if timers is not None:
        timers.start("forward_prepare_gaussians")
    raw = [getattr(pc, n, None) for n in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")]
    if not all(torch.is_tensor(t) for t in raw):
        raise TypeError("distributed_preprocess3dgs_and_all2all_final: `pc` must expose GaussianModel's raw parameters "
                        "(_xyz, _scaling, _rotation, _features_dc, _features_rest, _opacity; "
                        "scene/gaussian_model.py:219-242): the getters' activations run inside K1 / K11")
    if not all(t.is_cuda for t in raw):
        raise RuntimeError("distributed_preprocess3dgs_and_all2all_final: the model must live on the gfx950 device "
                           "(there is no CPU path)")
    # exp / normalize / sigmoid / cat of the getters (scene/gaussian_model.py:109-129) happen inside K1 / K11
    if timers is not None:
        timers.stop("forward_prepare_gaussians")
        timers.start("forward_preprocess_gaussians")

    rasterizers, cuda_args_list, params = [], [], []
    for camera, strategy in zip(batched_viewpoint_cameras, batched_strategies):
        ca = get_cuda_args_final(strategy, mode)
        ca["_gsr_timing"] = timing
        cuda_args_list.append(ca)
        settings = GaussianRasterizationSettings(
            image_height=int(camera.image_height),
            image_width=int(camera.image_width),
            tanfovx=math.tan(camera.FoVx * 0.5),
            tanfovy=math.tan(camera.FoVy * 0.5),
            bg=bg_color,
            scale_modifier=scaling_modifier,
            viewmatrix=camera.world_view_transform,
            projmatrix=camera.full_proj_transform,
            sh_degree=pc.active_sh_degree,
            campos=camera.camera_center,
            prefiltered=False,
            debug=pipe.debug,
        )
        rasterizers.append(GaussianRasterizer(raster_settings=settings))
    if len({(r.raster_settings.image_height, r.raster_settings.image_width) for r in rasterizers}) != 1:
        raise ValueError("all cameras of a batch must have one image size (the reference keeps ONE global image size, "
                         "utils/general_utils.py:89-93, set from the first training camera, scene/__init__.py:93-97)")
    # ONE launch for the whole batch: parameters read once, per-camera outputs camera-major
    packed = []
    for camera, rast in zip(batched_viewpoint_cameras, rasterizers):
        # the packed record is valid as long as the camera's tensors are the same objects with the same contents:
        # data_ptr + in-place version counter of each matrix (pose refinement / test-time edits repack)
        rs_k = rast.raster_settings
        key = (float(rs_k.tanfovx), float(rs_k.tanfovy)) + tuple(
            (t.data_ptr(), t._version) for t in (rs_k.viewmatrix, rs_k.projmatrix, rs_k.campos))
        cached = getattr(camera, "_gsr_packed", None)
        if cached is None or cached[0] != key or cached[1].device != raw[0].device:
            cached = (key, _dgr.pack_camera(rast.raster_settings))
            try:
                camera._gsr_packed = cached  # cameras are static: packed once
            except AttributeError:
                pass
        packed.append(cached[1])
    rs0 = rasterizers[0].raster_settings
    if len(packed) == 1:
        cams_block = packed[0].view(1, -1)
    elif _dgr.capturing() is not None:
        cams_block = torch.stack(packed)  # (a hipGraph capture: the records are slices of a block refreshed per replay)
    else:
        # the [B,40] block of a batch that has been seen before (the records themselves are cached per camera): the
        # stack is a launch plus ~50 us of host per iteration otherwise
        skey = tuple(id(t) for t in packed)
        hit = _CAMERA_BLOCKS.get(skey)
        if hit is not None and all(a is b and a._version == v for a, b, v in zip(hit[1], packed, hit[2])):
            cams_block = hit[0]
        else:
            cams_block = torch.stack(packed)
            if len(_CAMERA_BLOCKS) > 1024:
                _CAMERA_BLOCKS.clear()
            _CAMERA_BLOCKS[skey] = (cams_block, tuple(packed), tuple(t._version for t in packed))
    m2_all, rgb_all, co_all, radii_all, depths_all = _dgr.preprocess_gaussians_raw_batched(
        *raw, cams_block, pc.active_sh_degree,
        scaling_modifier, rs0.image_width, rs0.image_height, tanfov0=(rs0.tanfovx, rs0.tanfovy),
        cuda_args_list=cuda_args_list)
    for k in range(len(rasterizers)):
        means2D = m2_all[k]
        if mode == "train":
            _keep_grad_view(means2D)  # densification reads means2D.grad (scene/gaussian_model.py:1046-1052)
        params.append([means2D, rgb_all[k], co_all[k], radii_all[k], depths_all[k]])
    if timers is not None:
        timers.stop("forward_preprocess_gaussians")

    pkg = {
        "batched_locally_preprocessed_mean2D": [p[0] for p in params],
        # radii > 0 per camera (densification reads it, densification.py:16-26): computed when somebody looks
        "batched_locally_preprocessed_visibility_filter": _LazyPerCamera([p[3] for p in params], lambda r: r > 0),
        "batched_locally_preprocessed_radii": [p[3] for p in params],
        "batched_rasterizers": rasterizers,
        "batched_cuda_args": cuda_args_list,
    }
    W = utils.DEFAULT_GROUP.size()
    if W == 1 and not (_EXCHANGE_OPTIONS["forced"] and dist.is_initialized()):
        redistributed = tuple([p[c] for p in params] for c in range(5))
        sizes = [[[p[0].shape[0] for p in params]]]
        _fill(pkg, redistributed, sizes)
        return pkg
    if W * len(params) > 512:
        raise ValueError(f"world size x batch size = {W * len(params)} exceeds the 512 (destination, camera) segments "
                         "one exchange launch carries (include/gsraster.h: gsr_exchange_count)")
    if timers is not None:
        timers.start("forward_all_to_all_communication")
    mine = [k for k, st in enumerate(batched_strategies) if utils.GLOBAL_RANK in st.gpu_ids]

    def exchange(known=None):
        *redistributed, sizes, (events, token), pending = _batched_exchange_final(
            [p[0] for p in params], rgb_all, co_all, radii_all, depths_all, rasterizers, batched_strategies,
            speculate=False if _legacy else None, pipeline=False if _legacy else None, _known=known)
        _fill(pkg, redistributed, sizes)
        pkg["_exchange_events"] = events
        for ca in cuda_args_list:
            ca.pop("_exchange_token", None)
        if token is not None and mine:  # anchor of the exchange chain: the render op of the LAST camera this rank renders
            cuda_args_list[mine[-1]]["_exchange_token"] = token
        pkg["_exchange_token"] = token
        pkg.pop("_exchange_pending", None)
        if pending is not None:
            planner, chunkcnt, counts, lazy, staged = pending
            rendered = [(g, k) for k, st in enumerate(batched_strategies) for g in st.gpu_ids]

            def verify():
                """-> True when the speculative layout held (every count fitted its slab and no rendered band falls
                under the reference's < 10-Gaussian rule, which needs the true row count); otherwise the exchange has
                been repeated with exact sizes and the caller must render again.  The same decision on every rank:
                it is a function of the all-gathered matrix."""
                if pkg.get("_exchange_pending") is None:
                    return True
                pkg.pop("_exchange_pending")
                m, fitted = planner.resolve(staged)
                lazy._v = m.tolist()
                recv_rows = m.sum(axis=0).tolist()  # [destination][camera]: rows a band receives
                ok = fitted and all(recv_rows[g][k] >= 10 for (g, k) in rendered)
                planner.note(redone=not ok)
                if ok:
                    return True
                exchange_stats["redone"] += 1
                exchange(known=(chunkcnt, counts, lazy._v))
                return False

            lazy._resolver = verify
            pkg["_exchange_pending"] = verify

    exchange()
    if timers is not None:
        timers.stop("forward_all_to_all_communication")
    return pkg


def _fill(pkg, redistributed, sizes):
    for name, value in zip(("means2D", "rgb", "conic_opacity", "radii", "depths"), redistributed):
        pkg[f"batched_{name}_redistributed"] = value
    pkg["gpui_to_gpuj_imgk_size"] = sizes


def _render_cameras(pkg, batched_strategies):
    """-> (images, masks, late): `late` = the pair counts the render ops have NOT looked at yet (callables): every
    camera's K3-K7 and K8 are enqueued against the capacity of the kept sort scratch before the host reads the first
    count (diff_gaussian_rasterization.PendingPairs) -- round 6: the host used to stand between every camera's tile sort
    and its composite kernel (gaussian_renderer/__init__.py:1271-1282 is the reference's call site, whose rasterizer
    reads num_rendered back at the same place)."""
    timers = utils.get_timers()
    images, masks, late = [], [], []
    mine = [k for k, st in enumerate(batched_strategies) if utils.GLOBAL_RANK in st.gpu_ids]
    dev0 = pkg["batched_means2D_redistributed"][mine[0]].device if mine else None
    two = (_EXCHANGE_OPTIONS["camera_streams"] > 1 and len(mine) > 1 and dev0 is not None and dev0.type == "cuda"
           and _dgr.capturing() is None)
    if two:
        main = torch.cuda.current_stream(dev0)
        sides = [_side_stream(dev0, "camera0"), _side_stream(dev0, "camera1")]
        for s_ in sides:
            s_.wait_stream(main)  # K1's outputs (and the exchange's, whose events are waited for below)
    turn = 0
    for k, strategy in enumerate(batched_strategies):
        if utils.GLOBAL_RANK not in strategy.gpu_ids:
            images.append(None)
            masks.append(None)
            continue
        compute_locally = strategy.get_compute_locally()
        extended = strategy.get_extended_compute_locally()
        cuda_args = pkg["batched_cuda_args"][k]
        with (torch.cuda.stream(sides[turn & 1]) if two else _NULL):
            turn += 1
            ev = pkg.get("_exchange_events", None)
            if ev is not None and ev[k] is not None:
                _dgr.current_stream().wait_event(ev[k])  # camera k's exchange ran on the side stream
            means2D = pkg["batched_means2D_redistributed"][k]
            rgb = pkg["batched_rgb_redistributed"][k]
            conic_opacity = pkg["batched_conic_opacity_redistributed"][k]
            if timers is not None:
                timers.start("forward_render_gaussians")
            if means2D.shape[0] < 10:
                image = means2D.sum() + conic_opacity.sum() + rgb.sum()
                if cuda_args.get("_exchange_token") is not None:
                    image = image + 0.0 * cuda_args["_exchange_token"].sum()
                st = cuda_args["stats_collector"]
                st["forward_render_time"] = st["backward_render_time"] = st["forward_loss_time"] = 0.0
            else:
                rows_of = getattr(strategy, "_my_rows", None)  # host-known tile rows of this rank's band (the mask's rows)
                cuda_args["_gsr_band"] = rows_of() if rows_of is not None else None
                cuda_args["_gsr_pending"] = late
                try:
                    image, _, _, _ = pkg["batched_rasterizers"][k].render_gaussians(
                        means2D=means2D, conic_opacity=conic_opacity, rgb=rgb,
                        depths=pkg["batched_depths_redistributed"][k], radii=pkg["batched_radii_redistributed"][k],
                        compute_locally=compute_locally, extended_compute_locally=extended, cuda_args=cuda_args)
                finally:
                    cuda_args.pop("_gsr_pending", None)
            if two:
                image.record_stream(main)  # (allocated on the side stream, consumed by the loss on the caller's)
            if timers is not None:
                timers.stop("forward_render_gaussians")
        images.append(image)
        masks.append(compute_locally)
    if two:
        for s_ in sides:
            main.wait_stream(s_)
    return images, masks, late


def _settle(late):
    """look at the pair counts of the cameras just enqueued; a count that outgrew its capacity repeats that camera's
    tile sort and composite kernel in place (diff_gaussian_rasterization.PendingPairs)"""
    for settle in late:
        settle()
    del late[:]


settle = _settle  # public name: the counterpart of render_final(..., late=[...])


def render_final(batched_screenspace_pkg, batched_strategies, tile_size=16, late=None):
    """-> (images, masks) per camera: a [3,H,W] image (zero outside this rank's row band), a scalar
    stand-in when fewer than 10 Gaussians arrived (keeps the autograd graph and the exchange's
    backward alive, gaussian_renderer/__init__.py:1260-1269), or None when this rank renders no part
    of the camera.  A speculative exchange (capacity slabs) is verified HERE, after the renders have polled their pair
    counts -- the asynchronous copy of the exchange's counts is then long complete, so the check waits for nothing --
    and, had a slab overflowed, exchange and render are repeated with exact sizes.
    `late` (extension of this build, forward-only rendering on one rank): a list that receives the cameras' unsettled
    pair counts instead of having them settled here -- a render loop then enqueues view i + 1 before it calls
    gaussian_renderer.settle(late) for view i, whose image is FINAL only after that call (a view whose pair count
    outgrew the kept sort scratch is drawn again in place by it).  The reference's render driver consumes every image
    before it starts the next (render.py:87-95): leave `late` alone to get exactly that."""
    pkg = batched_screenspace_pkg
    images, masks, mine = _render_cameras(pkg, batched_strategies)
    if late is not None and pkg.get("_exchange_pending") is None:
        late.extend(mine)
        return images, masks
    late = mine
    _settle(late)
    verify = pkg.get("_exchange_pending")
    if verify is not None and not verify():
        images, masks, late = _render_cameras(pkg, batched_strategies)
        _settle(late)
    return images, masks


def _no_gsplat(*a, **k):
    raise NotImplementedError(
        "the gsplat backend is a third-party alternative whose source is not part of the reference tree "
        "(empty submodule); this build provides the default diff_gaussian_rasterization backend only")


gsplat_distributed_preprocess3dgs_and_all2all_final = _no_gsplat
gsplat_render_final = _no_gsplat


# ------------------------------------------------------------------ legacy single-camera surface
def _local_camera_index(batched_strategies):
    """the camera this rank renders: the reference indexes the batch with utils.DP_GROUP.rank()
    (gaussian_renderer/__init__.py:414-415); DP_GROUP is never assigned there, so fall back to the first
    camera whose strategy lists this rank"""
    dp = getattr(utils, "DP_GROUP", None)
    if dp is not None:
        return dp.rank()
    for k, strategy in enumerate(batched_strategies):
        if utils.GLOBAL_RANK in strategy.gpu_ids:
            return k
    return 0


def preprocess3dgs_and_all2all(batched_cameras, gaussians, pipe_args, background, batched_strategies, mode):
    """legacy entry point (gaussian_renderer/__init__.py:410-455): the package `render()` below consumes -- one
    local camera, keys `rasterizer`, `cuda_args`, `*_for_render` -- built from the live `final` path."""
    # one exchange for the whole batch on the current stream with exact sizes: render() below has neither the stream
    # hand-over nor the verification of the pipelined / speculative layouts, and without per-camera nodes there is no
    # token chain to anchor
    pkg = distributed_preprocess3dgs_and_all2all_final(batched_cameras, gaussians, pipe_args, background,
                                                       batched_strategies=batched_strategies, mode=mode, _legacy=True)
    k = _local_camera_index(batched_strategies)
    out = {
        "batched_locally_preprocessed_mean2D": pkg["batched_locally_preprocessed_mean2D"],
        "batched_locally_preprocessed_radii": pkg["batched_locally_preprocessed_radii"],
        "rasterizer": pkg["batched_rasterizers"][k],
        "cuda_args": pkg["batched_cuda_args"][k],
        "means2D_for_render": pkg["batched_means2D_redistributed"][k],
        "rgb_for_render": pkg["batched_rgb_redistributed"][k],
        "conic_opacity_for_render": pkg["batched_conic_opacity_redistributed"][k],
        "radii_for_render": pkg["batched_radii_redistributed"][k],
        "depths_for_render": pkg["batched_depths_redistributed"][k],
        "i2j_send_size": pkg["gpui_to_gpuj_imgk_size"],
    }
    if mode != "test":
        out["batched_locally_preprocessed_visibility_filter"] = pkg["batched_locally_preprocessed_visibility_filter"]
    return out


def render(screenspace_pkg, strategy=None):
    """legacy `gaussian_renderer.render()` (gaussian_renderer/__init__.py:458-507; named by north_star): one camera,
    -> (rendered_image, compute_locally).  Same scalar stand-in rule (< 1000 received Gaussians) and the same
    stats_collector keys as the reference's function."""
    timers = utils.get_timers()
    compute_locally = strategy.get_compute_locally()
    extended = strategy.get_extended_compute_locally()
    if timers is not None:
        timers.start("forward_render_gaussians")
    pkg = screenspace_pkg
    if pkg["means2D_for_render"].shape[0] < 1000:
        image = pkg["means2D_for_render"].sum() + pkg["conic_opacity_for_render"].sum() + pkg["rgb_for_render"].sum()
        st = pkg["cuda_args"]["stats_collector"]
        st["forward_render_time"] = st["backward_render_time"] = 0.0
        st["forward_loss_time"] = st["backward_loss_time"] = 0.0
        return image, compute_locally
    image, _, _, _ = pkg["rasterizer"].render_gaussians(
        means2D=pkg["means2D_for_render"], conic_opacity=pkg["conic_opacity_for_render"], rgb=pkg["rgb_for_render"],
        depths=pkg["depths_for_render"], radii=pkg["radii_for_render"], compute_locally=compute_locally,
        extended_compute_locally=extended, cuda_args=pkg["cuda_args"])
    if timers is not None:
        timers.stop("forward_render_gaussians")
    return image, compute_locally
