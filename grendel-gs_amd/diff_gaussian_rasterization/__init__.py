"""MI355X-native drop-in for the reference's `diff_gaussian_rasterization` operator module.

Host-side mirror (Python, because the reference's wrapper is Python) of the operator surface the
reference binds for its hot path -- same names, argument meaning and output order -- on top of the
C-ABI of include/gsraster.h (libgsraster.so, hand-written HIP for gfx950):

  GaussianRasterizationSettings        built at gaussian_renderer/__init__.py:930-943
  GaussianRasterizer.preprocess_gaussians   called at gaussian_renderer/__init__.py:949-956
  GaussianRasterizer.render_gaussians       called at gaussian_renderer/__init__.py:1271-1282
  _C.get_block_XY                      arguments/__init__.py:254-257
  _C.get_local2j_ids_bool              gaussian_renderer/workload_division.py:727-738
  load_image_tiles_by_pos / merge_image_tiles_by_pos / _C.get_touched_locally / ... : only dead
  callers in the reference (loss_distribution.py:136-213, workload_division.py:483) -> raise.

PyTorch is used for device memory, streams and autograd plumbing only; every kernel on the path is in
the HIP library and the module refuses to work without it (no CPU / eager fallback).
"""
import contextlib
import ctypes
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "_C", "load_image_tiles_by_pos",
           "merge_image_tiles_by_pos", "set_timing_mode", "fused_l1_ssim_band", "fused_band_loss", "fused_activations", "pack_camera",
           "preprocess_gaussians_raw_batched", "knn_mean_dist2", "group_rows", "gather_rows", "exchange_need",
           "exchange_count", "exchange_pack", "exchange_pack_slab", "exchange_unpack", "zeros_async", "scatter_add_rows", "densify_stats",
           "set_tie_order", "scatter_rows", "local_pixels", "GraphCapture", "capturing"]

BLOCK_X, BLOCK_Y, ONE_DIM_BLOCK_SIZE = 16, 16, 256

# how render_gaussians fills cuda_args["stats_collector"]["backward_render_time"]:
#   "sync"  : one HIP-event synchronisation at the end of the backward op -> exact per-call value
#   "stale" : no extra host sync; reports the most recent backward whose events have completed
#   "deferred": like "stale", and additionally leaves the HIP event pairs under the private keys
#             "_fwd_events" / "_bwd_events" so that a cooperating caller (this package's
#             gaussian_renderer.workload_division.finish_strategy_final) resolves exact values with the
#             one sync it performs anyway
#   "off"   : 0.0 (no events recorded)
_TIMING_MODE = "auto"
_last_backward_ms = 0.0


def set_timing_mode(mode):
    global _TIMING_MODE
    assert mode in ("auto", "sync", "stale", "deferred", "off")
    _TIMING_MODE = mode


def _timing_mode():
    if _TIMING_MODE != "auto":
        return _TIMING_MODE
    if torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size() > 1:
        return "sync"  # the load balancer consumes the value (workload_division.py:944-998)
    return "stale"


class _KernelTimer:
    """optional HIP-event brackets around each C-ABI call (bench.py's `roofline` leg): events are
    recorded on the stream the kernels are launched on and resolved by the caller after a sync.  Every record
    carries the launch's own sizes (N, P, D, Px, ...), so that bytes and time describe the SAME launches."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    class _Range:
        def __init__(self, owner, name, meta):
            self.owner, self.name, self.meta = owner, name, meta

        def __enter__(self):
            if self.owner.enabled:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.owner.enabled:
                self.e1.record()
                self.owner.records.setdefault(self.name, []).append((self.e0, self.e1, self.meta))

    def range(self, name, **meta):
        return _KernelTimer._Range(self, name, meta)

    def reset(self):
        self.records = {}

    def summary_ms(self):
        """name -> (launches, mean ms); call after torch.cuda.synchronize()"""
        return {k: (len(v), sum(r[0].elapsed_time(r[1]) for r in v) / len(v)) for k, v in self.records.items() if v}

    def launches(self):
        """name -> list of (ms, meta dict) per launch; call after torch.cuda.synchronize()"""
        return {k: [(r[0].elapsed_time(r[1]), r[2]) for r in v] for k, v in self.records.items() if v}


kernel_timer = _KernelTimer()



# ------------------------------------------------------------------ native GPU-timer logs (--zhx_time)
# The reference's rasterizer writes per-stage GPU times of the logged iterations to
# <log_folder>/gpu_time_ws=<W>_rk=<r>.log when cuda_args["zhx_time"] is "True" (gaussian_renderer/__init__.py:524-538);
# analyze_statistic.py:747-805 parses "it=<n>, ..." headers followed by "<stage name>: <ms> ms" lines and :1972-1991
# lists the stage names.  The same file is written here from HIP events around the C-ABI calls.  Stages this build has
# fused report 0: "24 ...updateTileTouched" = K3 + depth sort + offsets scan (the reference's 24 + 30), "50 SortPairs"
# = emission + tile sort + ranges (its 40 + 50 + 60).
_ZHX_STAGES = ["10 preprocess time", "24 updateDistributedStatLocally.updateTileTouched time", "30 InclusiveSum time",
               "40 duplicateWithKeys time", "50 SortPairs time", "60 identifyTileRanges time", "70 render time",
               "81 sum_n_render time", "82 sum_n_consider time", "83 sum_n_contrib time", "b10 render time",
               "b20 preprocess time"]
_NULL_RANGE = contextlib.nullcontext()


def _zhx_on(ca):
    if not isinstance(ca, dict) or str(ca.get("zhx_time")) != "True" or ca.get("mode") != "train":
        return False
    try:
        it, every = int(ca.get("iteration", -1)), max(int(ca.get("log_interval", 1)), 1)
    except (TypeError, ValueError):
        return False
    return bool(ca.get("log_folder")) and it >= 0 and (every == 1 or it % every == 1)


class _ZhxRange:
    def __init__(self, targets, label):
        self.targets, self.label = targets, label

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return self

    def __exit__(self, *exc):
        self.e1.record()
        for ca in self.targets:
            ca.setdefault("_zhx_events", []).append((self.label, self.e0, self.e1, 1.0 / len(self.targets)))
            if self.label == "b20 preprocess time":
                _zhx_flush(ca)


def zhx_range(cuda_args, label):
    """HIP-event bracket of one stage for every cuda_args dict (one, or a list for a camera batch) that asks for the
    native timer log on this iteration; a no-op context otherwise"""
    cas = cuda_args if isinstance(cuda_args, (list, tuple)) else (cuda_args,)
    targets = [ca for ca in cas if _zhx_on(ca)]
    return _ZhxRange(targets, label) if targets else _NULL_RANGE


def _zhx_flush(ca):
    events = ca.pop("_zhx_events", [])
    if not events:
        return
    events[-1][2].synchronize()
    ms = dict.fromkeys(_ZHX_STAGES, 0.0)
    for label, e0, e1, share in events:
        ms[label] = ms.get(label, 0.0) + e0.elapsed_time(e1) * share
    os.makedirs(ca["log_folder"], exist_ok=True)
    path = os.path.join(ca["log_folder"], f"gpu_time_ws={ca.get('world_size', '1')}_rk={ca.get('global_rank', '0')}.log")
    with open(path, "a") as f:
        f.write(f"it={ca.get('iteration')}, mode={ca.get('mode')}, mp_rank={ca.get('mp_rank', '0')}\n")
        for label in _ZHX_STAGES:
            f.write(f"{label}: {ms[label]:.6f} ms\n")


def composite_walked(reset=False):
    """-> (K8, K10) list entries walked since the last reset (include/gsraster.h: gsr_composite_walked); synchronous"""
    out = (ctypes.c_ulonglong * 2)()
    check(lib.gsr_composite_walked(out, 1 if reset else 0), "gsr_composite_walked")
    return int(out[0]), int(out[1])


def local_pixels(mask, W, H):
    """number of image pixels inside the tiles marked in `mask` (uint8/bool [TILE_Y*TILE_X]); host-syncing helper for
    bench bookkeeping, called after the timed region"""
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
    m = mask.reshape(gy, gx).to(torch.int64).cpu()
    hh = torch.full((gy,), BLOCK_Y, dtype=torch.int64)
    hh[-1] = H - (gy - 1) * BLOCK_Y
    ww = torch.full((gx,), BLOCK_X, dtype=torch.int64)
    ww[-1] = W - (gx - 1) * BLOCK_X
    return int((m * hh[:, None] * ww[None, :]).sum())


def _on(dev):
    """`with torch.cuda.device(dev)` only when `dev` is not already the current device (the guard costs ~10 us per use
    -- hipGetDevice / hipSetDevice pairs -- and every C-ABI call sits in one; one process drives one GPU)"""
    idx = dev.index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NULL_RANGE
    return torch.cuda.device(dev)


def _ptr(t):
    # a plain int: the prototypes declare c_void_p, ctypes converts (building a c_void_p object per argument was
    # ~60 objects per iteration)
    return t.data_ptr() if t is not None else None


def _stream():
    # the raw handle of the current stream of the current device (torch.cuda.current_stream() builds a Stream object
    # through three Python layers: ~10 us per call, and every C-ABI call needs one)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


_STREAM_OBJECTS = {}
_NO_STREAM_CACHE = os.environ.get("GSR_STREAM_CACHE", "1") == "0"  # env: A/B measurements only


def current_stream(dev=None):
    """torch.cuda.current_stream(dev), without its three Python layers per call (~5-9 us; an eager iteration of one rank
    asks a dozen times -- stream hand-overs, event records): the Stream object of a raw handle is built once"""
    idx = dev.index if (dev is not None and dev.index is not None) else torch._C._cuda_getDevice()
    if _NO_STREAM_CACHE:
        return torch.cuda.current_stream(idx)
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream(idx)
    return s


def _f32c(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"diff_gaussian_rasterization: `{name}` must live on the gfx950 device "
                           "(there is no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ------------------------------------------------------------------------------------ K1 / K11
def _grad_triple(g_means2D, g_rgb, g_conic_opacity, P, dev):
    """the three incoming gradients of a preprocess op as (means2D, rgb, conic_opacity, row stride in floats):
    when they are column views of row-major buffers with one common row stride (K10's [P,9] record) they are passed
    through untouched (stride > 0); otherwise dense copies / zeros (stride 0)."""
    gs = (g_means2D, g_rgb, g_conic_opacity)
    if all(g is not None and g.dtype == torch.float32 and g.dim() == 2 and g.shape[0] == P and
           (P <= 1 or (g.stride(1) == 1 and g.stride(0) == gs[0].stride(0))) for g in gs) and P > 1 and \
            gs[0].stride(0) > 4:
        return g_means2D, g_rgb, g_conic_opacity, int(gs[0].stride(0))

    def z(g, cols):
        return torch.zeros((P, cols), dtype=torch.float32, device=dev) if g is None else g.float().contiguous()

    return z(g_means2D, 2), z(g_rgb, 3), z(g_conic_opacity, 4), 0


class _PreprocessGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, scales, rotations, shs, opacities, raster_settings, cuda_args):
        rs = raster_settings
        means3D, scales, rotations = _f32c(means3D, "means3D"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
        shs, opacities = _f32c(shs, "shs"), _f32c(opacities, "opacities")
        P = means3D.shape[0]
        M = shs.shape[1] if shs.dim() == 3 else 0
        if shs.dim() != 3 or shs.shape[0] != P or shs.shape[2] != 3:
            raise ValueError(f"shs must be [P, K, 3], got {tuple(shs.shape)}")
        if scales.shape != (P, 3) or rotations.shape != (P, 4) or opacities.numel() != P:
            raise ValueError("scales [P,3], rotations [P,4], opacities [P,1] expected")
        dev = means3D.device
        view = _f32c(rs.viewmatrix, "viewmatrix")
        proj = _f32c(rs.projmatrix, "projmatrix")
        campos = _f32c(rs.campos, "campos")
        means2D = torch.empty((P, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((P,), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cov3D = torch.empty((P, 6), dtype=torch.float32, device=dev)
        conic_opacity = torch.empty((P, 4), dtype=torch.float32, device=dev)
        rgb = torch.empty((P, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((P, 3), dtype=torch.uint8, device=dev)
        ctx.cuda_args = cuda_args
        with _on(dev), kernel_timer.range("preprocess_forward", N=P, B=1, M=M), \
                zhx_range(cuda_args, "10 preprocess time"):
            check(lib.gsr_preprocess_forward(
                P, int(rs.sh_degree), M, _ptr(means3D), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(shs), _ptr(opacities), _ptr(view), _ptr(proj), _ptr(campos), int(rs.image_width),
                int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy), _ptr(means2D), _ptr(depths), _ptr(radii),
                _ptr(cov3D), _ptr(conic_opacity), _ptr(rgb), _ptr(clamped), _stream()), "gsr_preprocess_forward")
        ctx.raster_settings = rs
        ctx.M = M
        ctx.save_for_backward(means3D, scales, rotations, shs, view, proj, campos, radii, cov3D, clamped)
        ctx.mark_non_differentiable(radii, depths)
        return means2D, rgb, conic_opacity, radii, depths

    @staticmethod
    def backward(ctx, g_means2D, g_rgb, g_conic_opacity, g_radii, g_depths):
        rs = ctx.raster_settings
        means3D, scales, rotations, shs, view, proj, campos, radii, cov3D, clamped = ctx.saved_tensors
        P, M = means3D.shape[0], ctx.M
        dev = means3D.device

        g_means2D, g_rgb, g_conic_opacity, gstride = _grad_triple(g_means2D, g_rgb, g_conic_opacity, P, dev)
        d_means3D = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_scales = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_rot = torch.empty((P, 4), dtype=torch.float32, device=dev)
        d_shs = torch.empty((P, M, 3), dtype=torch.float32, device=dev)
        d_opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
        with _on(dev), kernel_timer.range("preprocess_backward", N=P, B=1, M=M), \
                zhx_range(ctx.cuda_args, "b20 preprocess time"):
            check(lib.gsr_preprocess_backward(
                P, int(rs.sh_degree), M, _ptr(means3D), _ptr(scales), float(rs.scale_modifier), _ptr(rotations),
                _ptr(shs), _ptr(view), _ptr(proj), _ptr(campos), int(rs.image_width), int(rs.image_height),
                float(rs.tanfovx), float(rs.tanfovy), _ptr(radii), _ptr(cov3D), _ptr(clamped), _ptr(g_means2D),
                _ptr(g_conic_opacity), _ptr(g_rgb), gstride, _ptr(d_means3D), _ptr(d_scales), _ptr(d_rot),
                _ptr(d_shs), _ptr(d_opac), _stream()), "gsr_preprocess_backward")
        return d_means3D, d_scales, d_rot, d_shs, d_opac, None, None


class _PreprocessGaussiansRaw(torch.autograd.Function):
    """K1 / K11 taking GaussianModel's RAW parameters; the getters' activations run inside the kernels
    (include/gsraster.h: gsr_preprocess_forward_raw / _backward_raw)."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, features_dc, features_rest, opacity, raster_settings, cuda_args):
        rs = raster_settings
        xyz, scaling, rotation = _f32c(xyz, "xyz"), _f32c(scaling, "scaling"), _f32c(rotation, "rotation")
        features_dc, features_rest = _f32c(features_dc, "features_dc"), _f32c(features_rest, "features_rest")
        opacity = _f32c(opacity, "opacity")
        P = xyz.shape[0]
        if features_dc.shape != (P, 1, 3) or features_rest.dim() != 3 or features_rest.shape[0] != P or \
                features_rest.shape[2] != 3 or features_rest.shape[1] < 1:
            raise ValueError("features_dc [P,1,3] and features_rest [P,K-1,3] expected")
        M = 1 + features_rest.shape[1]
        dev = xyz.device
        view, proj, campos = _f32c(rs.viewmatrix, "viewmatrix"), _f32c(rs.projmatrix, "projmatrix"), \
            _f32c(rs.campos, "campos")
        means2D = torch.empty((P, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((P,), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cov3D = torch.empty((P, 6), dtype=torch.float32, device=dev)
        conic_opacity = torch.empty((P, 4), dtype=torch.float32, device=dev)
        rgb = torch.empty((P, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((P, 3), dtype=torch.uint8, device=dev)
        ctx.cuda_args = cuda_args
        with _on(dev), kernel_timer.range("preprocess_forward", N=P, B=1, M=M), \
                zhx_range(cuda_args, "10 preprocess time"):
            check(lib.gsr_preprocess_forward_raw(
                P, int(rs.sh_degree), M, _ptr(xyz), _ptr(scaling), float(rs.scale_modifier), _ptr(rotation),
                _ptr(features_dc), _ptr(features_rest), _ptr(opacity), _ptr(view), _ptr(proj), _ptr(campos),
                int(rs.image_width), int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy), _ptr(means2D),
                _ptr(depths), _ptr(radii), _ptr(cov3D), _ptr(conic_opacity), _ptr(rgb), _ptr(clamped), _stream()),
                "gsr_preprocess_forward_raw")
        ctx.raster_settings, ctx.M = rs, M
        ctx.save_for_backward(xyz, scaling, rotation, features_dc, features_rest, opacity, view, proj, campos, radii,
                              cov3D, clamped)
        ctx.mark_non_differentiable(radii, depths)
        return means2D, rgb, conic_opacity, radii, depths

    @staticmethod
    def backward(ctx, g_means2D, g_rgb, g_conic_opacity, g_radii, g_depths):
        rs = ctx.raster_settings
        xyz, scaling, rotation, f_dc, f_rest, opacity, view, proj, campos, radii, cov3D, clamped = ctx.saved_tensors
        P, M = xyz.shape[0], ctx.M
        dev = xyz.device

        g_means2D, g_rgb, g_conic_opacity, gstride = _grad_triple(g_means2D, g_rgb, g_conic_opacity, P, dev)
        d_xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_scaling = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_rot = torch.empty((P, 4), dtype=torch.float32, device=dev)
        d_dc = torch.empty((P, 1, 3), dtype=torch.float32, device=dev)
        d_rest = torch.empty((P, M - 1, 3), dtype=torch.float32, device=dev)
        d_opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
        with _on(dev), kernel_timer.range("preprocess_backward", N=P, B=1, M=M), \
                zhx_range(ctx.cuda_args, "b20 preprocess time"):
            check(lib.gsr_preprocess_backward_raw(
                P, int(rs.sh_degree), M, _ptr(xyz), _ptr(scaling), float(rs.scale_modifier), _ptr(rotation),
                _ptr(f_dc), _ptr(f_rest), _ptr(opacity), _ptr(view), _ptr(proj), _ptr(campos), int(rs.image_width),
                int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy), _ptr(radii), _ptr(cov3D), _ptr(clamped),
                _ptr(g_means2D), _ptr(g_conic_opacity), _ptr(g_rgb), gstride, _ptr(d_xyz), _ptr(d_scaling),
                _ptr(d_rot), _ptr(d_dc), _ptr(d_rest), _ptr(d_opac), _stream()), "gsr_preprocess_backward_raw")
        return d_xyz, d_scaling, d_rot, d_dc, d_rest, d_opac, None, None


def pack_camera(raster_settings):
    """[40] float tensor { viewmatrix, projmatrix, campos, tanfovx, tanfovy, 0, 0, 0 } of one camera (device)"""
    rs = raster_settings
    dev = rs.viewmatrix.device
    tail = torch.tensor([float(rs.tanfovx), float(rs.tanfovy), 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
    return torch.cat([rs.viewmatrix.reshape(16).float(), rs.projmatrix.reshape(16).float(),
                      rs.campos.reshape(3).float(), tail]).contiguous()


def _batched_record(grads, B, P):
    """the [B*P, 9] fp32 record whose column blocks 0:2 / 2:5 / 5:9 of rows [kP, (k+1)P) ARE the gradients of camera
    k's means2D / rgb / conic_opacity, or None"""
    g0 = grads[0]
    if g0 is None or g0._base is None:
        return None
    base = g0._base
    if base.dtype != torch.float32 or not base.is_contiguous() or base.numel() != B * P * 9:
        return None
    p0 = base.data_ptr()
    for k in range(B):
        for col, (off, cols) in enumerate(((0, 2), (2, 3), (5, 4))):
            g = grads[5 * k + col]
            if g is None or g._base is not base or tuple(g.shape) != (P, cols) or g.stride() != (9, 1) or \
                    g.data_ptr() != p0 + 4 * (k * P * 9 + off):
                return None
    return base.view(B * P, 9)


# ---- deferred K11: the projection backward handed to an optimizer that fuses it with its step -------------------
# A sink (fused_optim.FusedAdam(fuse_backward=True)) registers itself here.  When the six raw parameters of a
# _PreprocessGaussiansRawBatched node are exactly the tensors the sink optimizes, the node's backward does not launch
# K11: it returns no gradients (`.grad` stays None) and hands the sink everything K11 needs; the sink's step() then
# runs K11 and Adam as ONE kernel (gsr_preprocess_backward_adam_raw_batched) -- or, whenever that is not possible,
# materializes the gradients with the plain K11 and steps as usual.  Opt-in: nothing is deferred without a sink.
_DEFERRED_SINK = [None]  # a weak reference: an optimizer that is dropped stops being the sink


def set_deferred_backward_sink(sink):
    """sink: object with accepts(params) -> bool and offer(PendingProjectionBackward), or None to switch deferral off.
    Held weakly -- a superseded optimizer that nobody references any more cannot swallow a backward."""
    import weakref

    _DEFERRED_SINK[0] = weakref.ref(sink) if sink is not None else None


def deferred_backward_sink():
    ref = _DEFERRED_SINK[0]
    return ref() if ref is not None else None


class PendingProjectionBackward:
    """K11's inputs of one backward pass, alive until the optimizer's step"""

    def __init__(self, params, cams, radii, cov3D, clamped, g_means2D, g_conic_opacity, g_rgb, gstride, meta, tanfov0):
        self.params = params  # xyz, scaling, rotation, features_dc, features_rest, opacity (the saved inputs)
        self.versions = tuple(t._version for t in params)
        self.cams, self.radii, self.cov3D, self.clamped = cams, radii, cov3D, clamped
        self.g_means2D, self.g_conic_opacity, self.g_rgb, self.gstride = g_means2D, g_conic_opacity, g_rgb, gstride
        self.meta, self.tanfov0 = meta, tanfov0
        self.stream = current_stream(params[0].device) if params[0].is_cuda else None

    def join_stream(self):
        """make the current stream wait for the stream the backward ran on (no-op when they are the same)"""
        if self.stream is not None:
            cur = current_stream(self.params[0].device)
            if cur != self.stream:
                cur.wait_stream(self.stream)

    def materialize(self):
        """-> the six gradients, computed by the plain K11 (what the node's backward would have returned)"""
        self.join_stream()
        return _launch_k11(self.params, self.cams, self.radii, self.cov3D, self.clamped, self.g_means2D,
                           self.g_conic_opacity, self.g_rgb, self.gstride, self.meta, self.tanfov0, None)

    def fused_step(self, exp_avgs, exp_avg_sqs, lrs, beta1s, beta2s, epss, steps, grad_scale, cache=None):
        """K11 + Adam of the six tensors in one launch; arguments in the order of self.params.  `cache`: a dict the
        caller keeps between steps -- the ctypes tables of the moments' addresses and of the betas / eps only change
        when the model is rebuilt, and this call sits on the host's critical path in front of the launch."""
        self.join_stream()
        xyz, scaling, rotation, f_dc, f_rest, opacity = self.params
        deg, smod, W, H, M = self.meta
        P, B = xyz.shape[0], self.cams.shape[0]
        VP, D, I64 = ctypes.c_void_p * 6, ctypes.c_double * 6, ctypes.c_int64 * 6
        key = (tuple(t.data_ptr() for t in exp_avgs), tuple(t.data_ptr() for t in exp_avg_sqs), tuple(beta1s),
               tuple(beta2s), tuple(epss))
        tabs = cache.get("tabs") if cache is not None else None
        if tabs is None or tabs[0] != key:
            tabs = (key, VP(*key[0]), VP(*key[1]), D(*beta1s), D(*beta2s), D(*epss))
            if cache is not None:
                cache["tabs"] = tabs
        tf = (ctypes.c_float * 2)(float(self.tanfov0[0]), float(self.tanfov0[1])) \
            if (B == 1 and self.tanfov0 is not None) else None
        cap_ctx = _CAPTURE[0]
        if cap_ctx is not None:
            # captured launch: lr / bias corrections come from the capture's device block at execution time, and the
            # launch is a no-op when a capacity check of the same replay has raised the flag word
            with _on(xyz.device):
                check(lib.gsr_preprocess_backward_adam_raw_batched_dyn(
                    P, B, deg, M, _ptr(xyz), _ptr(scaling), smod, _ptr(rotation), _ptr(f_dc), _ptr(f_rest),
                    _ptr(opacity), _ptr(self.cams), W, H, _ptr(self.radii), _ptr(self.cov3D), _ptr(self.clamped),
                    _ptr(self.g_means2D), _ptr(self.g_conic_opacity), _ptr(self.g_rgb), self.gstride, tabs[1], tabs[2],
                    None, tabs[3], tabs[4], tabs[5], None, float(grad_scale), tf, _ptr(cap_ctx.dyn), _ptr(cap_ctx.flag),
                    _stream()), "gsr_preprocess_backward_adam_raw_batched_dyn")
            return
        with _on(xyz.device), kernel_timer.range("preprocess_backward_adam", N=P, B=B, M=M):
            check(lib.gsr_preprocess_backward_adam_raw_batched(
                P, B, deg, M, _ptr(xyz), _ptr(scaling), smod, _ptr(rotation), _ptr(f_dc), _ptr(f_rest), _ptr(opacity),
                _ptr(self.cams), W, H, _ptr(self.radii), _ptr(self.cov3D), _ptr(self.clamped), _ptr(self.g_means2D),
                _ptr(self.g_conic_opacity), _ptr(self.g_rgb), self.gstride, tabs[1], tabs[2], D(*lrs), tabs[3], tabs[4],
                tabs[5], I64(*steps), float(grad_scale), tf, _stream()), "gsr_preprocess_backward_adam_raw_batched")


def _launch_k11(params, cams, radii, cov3D, clamped, g_means2D, g_conic_opacity, g_rgb, gstride, meta, tanfov0,
                cuda_args_list):
    xyz, scaling, rotation, f_dc, f_rest, opacity = params
    deg, smod, W, H, M = meta
    P, B = xyz.shape[0], cams.shape[0]
    dev = xyz.device
    d_xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
    d_scaling = torch.empty((P, 3), dtype=torch.float32, device=dev)
    d_rot = torch.empty((P, 4), dtype=torch.float32, device=dev)
    d_dc = torch.empty((P, 1, 3), dtype=torch.float32, device=dev)
    d_rest = torch.empty((P, M - 1, 3), dtype=torch.float32, device=dev)
    d_opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
    with _on(dev), kernel_timer.range("preprocess_backward", N=P, B=B, M=M), \
            zhx_range(cuda_args_list, "b20 preprocess time"):
        if B == 1 and tanfov0 is not None:
            # single camera: the leaner one-camera kernel (no accumulators); camera fields are slices of `cams`
            base = cams.data_ptr()
            check(lib.gsr_preprocess_backward_raw(
                P, deg, M, _ptr(xyz), _ptr(scaling), smod, _ptr(rotation), _ptr(f_dc), _ptr(f_rest),
                _ptr(opacity), ctypes.c_void_p(base), ctypes.c_void_p(base + 64), ctypes.c_void_p(base + 128), W,
                H, float(tanfov0[0]), float(tanfov0[1]), _ptr(radii), _ptr(cov3D), _ptr(clamped),
                _ptr(g_means2D), _ptr(g_conic_opacity), _ptr(g_rgb), gstride, _ptr(d_xyz), _ptr(d_scaling),
                _ptr(d_rot), _ptr(d_dc), _ptr(d_rest), _ptr(d_opac), _stream()), "gsr_preprocess_backward_raw")
        else:
            check(lib.gsr_preprocess_backward_raw_batched(
                P, B, deg, M, _ptr(xyz), _ptr(scaling), smod, _ptr(rotation), _ptr(f_dc), _ptr(f_rest),
                _ptr(opacity), _ptr(cams), W, H, _ptr(radii), _ptr(cov3D), _ptr(clamped), _ptr(g_means2D),
                _ptr(g_conic_opacity), _ptr(g_rgb), gstride, _ptr(d_xyz), _ptr(d_scaling), _ptr(d_rot),
                _ptr(d_dc), _ptr(d_rest), _ptr(d_opac), _stream()), "gsr_preprocess_backward_raw_batched")
    return d_xyz, d_scaling, d_rot, d_dc, d_rest, d_opac


class _PreprocessGaussiansRawBatched(torch.autograd.Function):
    """K1 / K11 for a batch of B cameras in one launch each way (gsr_preprocess_*_raw_batched)."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, features_dc, features_rest, opacity, cams, sh_degree, scale_modifier,
                width, height, tanfov0, cuda_args_list=None):
        xyz, scaling, rotation = _f32c(xyz, "xyz"), _f32c(scaling, "scaling"), _f32c(rotation, "rotation")
        features_dc, features_rest = _f32c(features_dc, "features_dc"), _f32c(features_rest, "features_rest")
        opacity, cams = _f32c(opacity, "opacity"), _f32c(cams, "cams")
        ctx.tanfov0 = tanfov0
        ctx.cuda_args_list = cuda_args_list
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero-filled tensors
        P, B = xyz.shape[0], cams.shape[0]
        if cams.dim() != 2 or cams.shape[1] != 40:
            raise ValueError("cams must be [B, 40]")
        M = 1 + features_rest.shape[1]
        dev = xyz.device
        means2D = torch.empty((B, P, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((B, P), dtype=torch.float32, device=dev)
        radii = torch.empty((B, P), dtype=torch.int32, device=dev)
        cov3D = torch.empty((P, 6), dtype=torch.float32, device=dev)
        conic_opacity = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        rgb = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
        clamped = torch.empty((B, P, 3), dtype=torch.uint8, device=dev)
        with _on(dev), kernel_timer.range("preprocess_forward", N=P, B=B, M=M), \
                zhx_range(cuda_args_list, "10 preprocess time"):
            check(lib.gsr_preprocess_forward_raw_batched(
                P, B, int(sh_degree), M, _ptr(xyz), _ptr(scaling), float(scale_modifier), _ptr(rotation),
                _ptr(features_dc), _ptr(features_rest), _ptr(opacity), _ptr(cams), int(width), int(height),
                _ptr(means2D), _ptr(depths), _ptr(radii), _ptr(cov3D), _ptr(conic_opacity), _ptr(rgb), _ptr(clamped),
                _stream()), "gsr_preprocess_forward_raw_batched")
        ctx.meta = (int(sh_degree), float(scale_modifier), int(width), int(height), M)
        ctx.save_for_backward(xyz, scaling, rotation, features_dc, features_rest, opacity, cams, radii, cov3D, clamped)
        # per-camera outputs are separate autograd outputs (dense views of the camera-major buffers): a caller
        # that uses camera k alone gets its gradient straight back, without the zeros + copy of a select-backward
        outs = []
        for k in range(B):
            outs += [means2D[k], rgb[k], conic_opacity[k], radii[k], depths[k]]
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i % 5 >= 3])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        xyz, scaling, rotation, f_dc, f_rest, opacity, cams, radii, cov3D, clamped = ctx.saved_tensors
        deg, smod, W, H, M = ctx.meta
        P, B = xyz.shape[0], cams.shape[0]
        dev = xyz.device

        def assemble(col, cols):
            """[B,P,cols] gradient buffer of output `col` (0 means2D, 1 rgb, 2 conic_opacity) without copies when
            there is one camera or the per-camera gradients already are the slices of one dense buffer"""
            gs = [grads[5 * k + col] for k in range(B)]
            if B == 1:
                return torch.zeros((1, P, cols), dtype=torch.float32, device=dev) if gs[0] is None \
                    else gs[0].float().contiguous()
            if all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for g in gs):
                base, step = gs[0]._base, P * cols * 4
                if base is not None and all(g._base is base and g.data_ptr() == gs[0].data_ptr() + k * step
                                            for k, g in enumerate(gs)):
                    return gs[0]  # slices of one dense [B,P,cols] block (e.g. an unbound stack gradient)
            out = torch.zeros((B, P, cols), dtype=torch.float32, device=dev)
            for k, g in enumerate(gs):
                if g is not None:
                    out[k].copy_(g)
            return out

        if B == 1:  # K10's [P,9] record (or any common-stride column views) goes straight into the kernel
            g_means2D, g_rgb, g_conic_opacity, gstride = _grad_triple(grads[0], grads[1], grads[2], P, dev)
        else:
            rec = _batched_record(grads, B, P)
            if rec is not None:  # column views of ONE [B,P,9] record (the exchange's backward): read through the stride
                g_means2D, g_rgb, g_conic_opacity, gstride = rec[:, 0:2], rec[:, 2:5], rec[:, 5:9], 9
            else:
                g_means2D, g_rgb, g_conic_opacity, gstride = assemble(0, 2), assemble(1, 3), assemble(2, 4), 0
        params = (xyz, scaling, rotation, f_dc, f_rest, opacity)
        sink = deferred_backward_sink()
        if sink is not None and M == 16 and sink.accepts(params):
            # K11 runs inside the optimizer's step (fused with Adam); `.grad` of the six parameters stays None
            sink.offer(PendingProjectionBackward(params, cams, radii, cov3D, clamped, g_means2D, g_conic_opacity,
                                                 g_rgb, gstride, ctx.meta, ctx.tanfov0))
            return (None,) * 13
        return _launch_k11(params, cams, radii, cov3D, clamped, g_means2D, g_conic_opacity, g_rgb, gstride, ctx.meta,
                           ctx.tanfov0, ctx.cuda_args_list) + (None,) * 7


def preprocess_gaussians_raw_batched(xyz, scaling, rotation, features_dc, features_rest, opacity, cams, sh_degree,
                                     scale_modifier, width, height, tanfov0=None, cuda_args_list=None):
    """-> per-camera lists (means2D [P,2], rgb [P,3], conic_opacity [P,4], radii int32 [P], depths [P]) x B for the B
    cameras packed in `cams` [B,40] (pack_camera); entry k of each list is a dense view of a camera-major [B,P,.]
    buffer.  Extension of this build used by the gaussian_renderer mirror."""
    flat = _PreprocessGaussiansRawBatched.apply(xyz, scaling, rotation, features_dc, features_rest, opacity, cams,
                                                sh_degree, scale_modifier, width, height, tanfov0, cuda_args_list)
    return tuple([flat[5 * k + c] for k in range(cams.shape[0])] for c in range(5))


# ------------------------------------------------------------------------------- K3..K8 / K10
def _bucket(nbytes):
    """Round a D-dependent buffer size up to 1/8 of its leading power of two.  The pair count differs from view to
    view, and every new size is a miss of torch's caching allocator -- a hipMalloc (tens of ms for the multi-GB
    buffers of a 4K / 40 M-Gaussian view) inside the iteration; with <= 12.5 % of slack a handful of buckets
    serves every view."""
    if nbytes < (1 << 20):
        return nbytes
    step = 1 << (int(nbytes).bit_length() - 4)
    return (nbytes + step - 1) // step * step


_SORT_SCRATCH = {}  # (device index, stream) -> grow-only uint8 buffer
_MAX_PAIRS = {}     # (device index, stream) -> largest pair count sorted there


def _sort_scratch(nbytes, dev):
    """The sort's ping-pong buffers are dead when gsr_bin_sort's kernels have run, so consecutive calls on one stream
    can share ONE grow-only buffer (stream order makes the reuse safe; another stream gets its own).  Measured on the
    40 M-Gaussian / 4K shape: 260 -> ~30 ms of 'binning' per view were allocator misses on the 22 GB scratch."""
    idx = dev.index if dev.index is not None else torch._C._cuda_getDevice()
    key = (idx, int(torch._C._cuda_getCurrentRawStream(idx)))
    buf = _SORT_SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        _SORT_SCRATCH.pop(key, None)
        buf = None  # drop the old block before asking for the larger one
        buf = torch.empty((_bucket(nbytes),), dtype=torch.uint8, device=dev)
        _SORT_SCRATCH[key] = buf
    return buf


def release_workspaces():
    """drop the cached sort scratch (e.g. before switching to a much smaller scene)"""
    _SORT_SCRATCH.clear()
    _MAX_PAIRS.clear()


_SPECULATIVE_SORT = [os.environ.get("GSR_SPECULATIVE_SORT", "1") != "0"]  # env: A/B measurements only
_SEGMENTS = [{"0": False, "always": "always"}.get(os.environ.get("GSR_SEGMENTS", "1"), True)]  # env: A/B only
_BAND_GRID = [os.environ.get("GSR_BAND_GRID", "1") != "0"]  # env: A/B measurements only
_ZERO_IN_FORWARD = [os.environ.get("GSR_ZERO_IN_FORWARD", "1") != "0"]  # env: A/B measurements only


def set_list_segments(on):
    """True (default): on THIN bands (<= 2048 tiles, named by the caller through cuda_args["_gsr_band"]) the composite
    forward leaves checkpoints every 256 walked list entries and the backward runs the segments in parallel workgroups
    (gsr_render_*_seg); "always": on every launch; False: one workgroup walks a tile's whole list"""
    _SEGMENTS[0] = "always" if on == "always" else bool(on)


def set_speculative_sort(on):
    """True (default): launch the tile sort for the scratch's capacity before the pair count has reached the host
    (gsr_bin_sort_bounded); False: the reference's order -- read num_rendered, size the buffers, sort."""
    _SPECULATIVE_SORT[0] = bool(on)


GSR_ERETRY = -3


_PERSIST_USER = [None]  # the mode a caller chose explicitly (None: the environment's)
_LAST_SEGMENTS = [0]    # row segments of the last view binned while kernel_timer was enabled


def set_bin_persistent(mode, _internal=False):
    """K3-K7 as two persistent launches with grid-wide barriers (default) or as the nine launches of the look-back
    pipeline: "env" (GSR_BIN_PERSIST, default on), False / "off", "prepare", "sort", True / "both"
    (include/gsraster.h: gsr_set_bin_persistent).  Lists are bit-identical either way.  A choice made through this
    function is the caller's: the package's own temporary switches (gaussian_renderer: look-back pipeline while an
    exchange overlaps on the side stream) never override it."""
    code = {"env": -1, False: 0, "off": 0, "prepare": 1, "sort": 2, True: 3, "both": 3}[mode]
    if not _internal:
        _PERSIST_USER[0] = None if mode == "env" else mode
    check(lib.gsr_set_bin_persistent(code), "gsr_set_bin_persistent")


def bin_persistent_user_choice():
    """the mode set_bin_persistent was last given by a caller, or None (the environment decides)"""
    return _PERSIST_USER[0]


def set_bin_rowmajor(mode):
    """K5-K7 with one pass over the pairs (csrc/binning_rows.h; include/gsraster.h: gsr_set_bin_rowmajor): "env"
    (GSR_BIN_ROWS, default on), True / "on", False / "off" (the two-pass split-key pipelines).  Same lists either way."""
    code = {"env": -1, False: 0, "off": 0, True: 1, "on": 1}[mode]
    check(lib.gsr_set_bin_rowmajor(code), "gsr_set_bin_rowmajor")


def bin_persist_status():
    """-> dict(done, fault_code, faults, solo_recoveries) of the persistent binning launches on the current device
    (include/gsraster.h: gsr_bin_persist_status).  `solo_recoveries` counts the views whose tile sort was finished by one
    workgroup because the grid did not become resident within the time-out (a shared device); `faults` the barrier
    time-outs that the next binning call reports as an exception (GSR_EFAULT)."""
    w = (ctypes.c_uint32 * 4)()
    check(lib.gsr_bin_persist_status(w), "gsr_bin_persist_status")
    return dict(done=int(w[0]), fault_code=int(w[1]), faults=int(w[2]), solo_recoveries=int(w[3]))


def set_tile_cull(mode):
    """exact tile culling in K3 (include/gsraster.h: gsr_set_tile_cull): "env" (GSR_TILE_CULL, default off), False / "off",
    True / "on", "auto" (on for frames of more than 16384 tiles, i.e. above ~2048 x 2048).  Lists stay order-preserving
    subsequences of the uncut ones; image and gradients unchanged up to the blend's summation order."""
    code = {"env": -1, False: 0, "off": 0, True: 1, "on": 1, "auto": 2}[mode]
    check(lib.gsr_set_tile_cull(code), "gsr_set_tile_cull")


def set_tie_order(order):
    """order of Gaussians with EXACTLY equal depth inside a tile list: "arrival" (default; the reference: index in the
    arrays the op is given, i.e. (source rank, index on the source) at world size > 1) or "position" (means2D.x, .y,
    then index): the image and the gradients then do not depend on the number of ranks or on the order of the
    Gaussians (include/gsraster.h: gsr_set_depth_tie_order)."""
    if order not in ("arrival", "position"):
        raise ValueError('tie order must be "arrival" or "position"')
    check(lib.gsr_set_depth_tie_order(1 if order == "position" else 0), "gsr_set_depth_tie_order")


# ---- an iteration captured in a hipGraph (graphed_step.GraphedIteration) --------------------------------------------
# Inside a capture nothing may reach the host: the pair count is not polled, buffers are sized from the counts of earlier
# (eager) iterations, and the kernels whose capacity may not hold raise bits of the capture's device flag word
# (include/gsraster.h: gsr_flag_if_greater), which turns the optimizer launch of the same replay into a no-op.
FLAG_PAIRS, FLAG_SLAB, FLAG_FEW = 1, 2, 4
_CAPTURE = [None]


class GraphCapture:
    """what the ops need to know while an iteration is being captured (set by graphed_step.GraphedIteration)"""

    def __init__(self, flag, dyn, pair_slack=1.25):
        self.flag, self.dyn, self.pair_slack = flag, dyn, pair_slack  # device uint32 [1], device float32 [12]
        self.pairs = []     # per render op of the iteration: (pinned int32 [1] that receives D, capacity)
        self.counts = None  # (pinned int32 [W*W*B] that receives the exchange's all-gathered counts, caps numpy, shape)
        self.slab_caps_dev = None  # device int32 [W*W*B]: the exchange planner's capacities, uploaded before the capture
        # pinned memory is allocated BEFORE the capture starts (hipHostMalloc is not a capturable call)
        self._words, self._next = torch.zeros((8192,), dtype=torch.int32).pin_memory(), 0
        # device timestamps for the load balancer (include/gsraster.h: gsr_stamp): (pinned int64 ring, slots, words per
        # slot), set by GraphedIteration(timings=True); `stamps` lists what was taken, in launch order, as (kind, tag)
        self.stamp_ring = None
        self.stamps = []

    def stamp(self, kind, tag):
        """a device timestamp at this point of the captured stream (kind: "fwd0" / "fwd1" around K3-K8, "loss0" /
        "loss1" around the loss forward, "bwd0" / "bwd1" around K10; tag: the camera's stats_collector)"""
        if self.stamp_ring is None:
            return
        ring, slots, per = self.stamp_ring
        idx = len(self.stamps)
        if idx >= per:
            raise RuntimeError("graph capture: out of timestamp words")
        check(lib.gsr_stamp(self.dyn.data_ptr() + 48, ring.data_ptr(), slots, per, idx, _stream()), "gsr_stamp")
        self.stamps.append((kind, tag))

    def take_pinned(self, n):
        """-> a pinned int32 view of n words (from the block allocated before the capture)"""
        if self._next + n > self._words.numel():
            raise RuntimeError("graph capture: out of pinned result words")
        v = self._words[self._next:self._next + n]
        self._next += n
        return v

    def pair_capacity(self, dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        seen = max([v for (d, _s), v in _MAX_PAIRS.items() if d == idx], default=0)
        if seen <= 0:
            raise RuntimeError("graph capture: no eager iteration has sorted a view on this device yet")
        return _bucket(4 * int(seen * self.pair_slack + 4096)) // 4


def capturing():
    return _CAPTURE[0]


_DEBUG_RANGES = []  # (GSR_GRAPH_DEBUG=1) buffers of captured iterations: eager allocations must never overlap them


def _debug_overlap(name, t):
    a0, a1 = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
    for n, p, sz in _DEBUG_RANGES:
        if a0 < p + sz and p < a1:
            print(f"[graphed] OVERLAP: eager {name} [{a0:#x}, {a1:#x}) with captured {n} [{p:#x}, {p + sz:#x})",
                  file=__import__("sys").stderr, flush=True)


def _bin_gaussians_captured(cap_ctx, means2D, depths, radii, conic_opacity, compute_locally, width, height):
    """K3-K7 inside a hipGraph capture: the tile sort is launched for a CAPACITY taken from earlier iterations, the pair
    count stays on the device (copied to a pinned word the host reads after the replay) and a count above the capacity
    raises FLAG_PAIRS instead of a re-sort.  Returns the capacity as D."""
    P = means2D.shape[0]
    dev = means2D.device
    gx, gy = (width + BLOCK_X - 1) // BLOCK_X, (height + BLOCK_Y - 1) // BLOCK_Y
    ranges = torch.empty((gx * gy + 1, 2), dtype=torch.int32, device=dev)[:gx * gy]
    prep_bytes = lib.gsr_bin_prepare_bytes(P, width, height)
    prep = torch.empty((max(prep_bytes, 4),), dtype=torch.uint8, device=dev)
    stream = _stream()
    ticket = ctypes.c_uint32(0)
    check(lib.gsr_bin_prepare_async(P, width, height, _ptr(means2D), _ptr(depths), _ptr(radii), _ptr(conic_opacity),
                                    _ptr(compute_locally), _ptr(prep), prep_bytes, ctypes.byref(ticket), stream),
          "gsr_bin_prepare_async")
    cap = cap_ctx.pair_capacity(dev)
    if int(lib.gsr_bin_sort_capacity(P, lib.gsr_bin_sort_bytes(P, cap, width, height), width, height)) < cap:
        raise RuntimeError("graph capture: frames above 256 x 256 tiles have no bounded tile sort")
    sort_bytes = lib.gsr_bin_sort_bytes(P, cap, width, height)
    scratch = torch.empty((max(sort_bytes, 4),), dtype=torch.uint8, device=dev)
    point_list = torch.empty((cap,), dtype=torch.int32, device=dev)
    check(lib.gsr_bin_sort_bounded(P, width, height, _ptr(compute_locally), _ptr(prep), cap, _ptr(scratch), sort_bytes,
                                   _ptr(point_list), _ptr(ranges), stream), "gsr_bin_sort_bounded")
    off = int(lib.gsr_bin_total_offset(P, width, height))
    host = cap_ctx.take_pinned(1)  # the kernel leaves the pair count there (pinned, device-accessible)
    check(lib.gsr_flag_if_greater(prep.data_ptr() + off, cap, _ptr(cap_ctx.flag), FLAG_PAIRS, host.data_ptr(), stream),
          "gsr_flag_if_greater")
    cap_ctx.pairs.append((host, cap))
    if os.environ.get("GSR_GRAPH_DEBUG") == "1":
        _DEBUG_RANGES.extend([("prep", prep.data_ptr(), prep.numel()), ("scratch", scratch.data_ptr(), scratch.numel()),
                              ("point_list", point_list.data_ptr(), 4 * point_list.numel())])
    return point_list, ranges, cap


class PendingPairs:
    """The pair count of a view whose K3-K7 (and, by the time finish() is called, K8) are already in flight (round 6).
    The kernels that consume the lists read the range table, never the count, so the host need not stand between the
    tile sort and the composite kernel: it launches both against the CAPACITY of the kept scratch and looks at the
    count afterwards.  finish() -> (D, new_point_list or None): None when the count fitted the capacity (the lists in
    flight are complete); otherwise the bounded sort wrote nothing and left every range empty (the composite kernel drew
    the background), the tile sort has been repeated HERE with exact sizes into a new point_list, and the caller
    launches its consumer again (same outputs, same stream: stream order makes the repeat invisible to later work)."""

    def __init__(self, ticket, sorted_, cap, P, width, height, mask, prep, ranges, dev, stream, key, cuda_args):
        self.ticket, self.sorted, self.cap = ticket, sorted_, cap
        self.P, self.width, self.height = P, width, height
        self.mask, self.prep, self.ranges = mask, prep, ranges  # (keeps the prepare workspace alive for a repeat)
        self.dev, self.stream, self.key, self.cuda_args = dev, stream, key, cuda_args
        self.D = None

    def finish(self):
        # (the stream guard only when the caller settles from another stream: entering / leaving it is ~15 us of Python)
        same = (not _NO_STREAM_CACHE) and self.stream is current_stream(self.dev)
        with _on(self.dev), (_NULL_RANGE if same else torch.cuda.stream(self.stream)):
            stream = _stream()
            D = ctypes.c_int64(0)
            rc = lib.gsr_bin_count_wait(self.ticket, ctypes.byref(D), stream)
            if rc != GSR_ERETRY:
                check(rc, "gsr_bin_count_wait")
            self.D = D = int(D.value)
            if D > _MAX_PAIRS.get(self.key, 0):
                _MAX_PAIRS[self.key] = D
            if rc == 0 and self.sorted and D <= self.cap:
                self.prep = None
                return D, None
            sort_bytes = lib.gsr_bin_sort_bytes(self.P, D, self.width, self.height)
            scratch = _sort_scratch(max(sort_bytes, 4), self.dev)
            point_list = torch.empty((_bucket(max(D, 1) * 4) // 4,), dtype=torch.int32, device=self.dev)[:max(D, 1)]
            with zhx_range(self.cuda_args, "50 SortPairs time"):
                check(lib.gsr_bin_sort(self.P, self.width, self.height, _ptr(self.mask), _ptr(self.prep), D,
                                       _ptr(scratch), sort_bytes, _ptr(point_list), _ptr(self.ranges), stream),
                      "gsr_bin_sort")
            self.prep = None
            return D, point_list


def bin_gaussians(means2D, depths, radii, conic_opacity, compute_locally, width, height, cuda_args=None, defer=False):
    """K3-K7: returns (point_list uint32-as-int32 [D], ranges int32 [tiles,2], D).  One host read-back (the pair count
    D, like the reference's own num_rendered) -- but the GPU does not wait for it: once a view of this size has been
    sorted, the next sort is launched for the CAPACITY of the scratch kept from then, the kernels take D from device
    memory, and the host reads D while they run (it only re-sorts if D outgrew the capacity).
    `defer` (round 6): do not even read D here -- returns (point_list [capacity], ranges, PendingPairs) and the caller
    calls .finish() after it has launched the consumer of the lists.
    Per tile the list is the reference's (depth, then index) order restricted to the Gaussians that
    can reach alpha >= 1/255 somewhere in the tile's neighbourhood (see include/gsraster.h)."""
    if _CAPTURE[0] is not None:
        return _bin_gaussians_captured(_CAPTURE[0], means2D, depths, radii, conic_opacity, compute_locally, width, height)
    P = means2D.shape[0]
    dev = means2D.device
    gx, gy = (width + BLOCK_X - 1) // BLOCK_X, (height + BLOCK_Y - 1) // BLOCK_Y
    # tiles + 1 rows: the last one receives the row hull of the mask (the band), which K8 / K10 read through the same
    # pointer; callers see the per-tile rows
    ranges = torch.empty((gx * gy + 1, 2), dtype=torch.int32, device=dev)[:gx * gy]
    prep_bytes = lib.gsr_bin_prepare_bytes(P, width, height)
    prep = torch.empty((max(prep_bytes, 4),), dtype=torch.uint8, device=dev)
    if _DEBUG_RANGES:
        _debug_overlap("prep", prep)
        _debug_overlap("means2D", means2D)
    stream = _stream()
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(stream.value or 0))
    cap, point_list = 0, None
    kept = _SORT_SCRATCH.get(key)
    if _SPECULATIVE_SORT[0] and kept is not None:
        # pairs the kept scratch can sort, but no more than the largest count seen here plus some slack (point_list
        # is allocated at the capacity)
        cap = min(int(lib.gsr_bin_sort_capacity(P, kept.numel(), width, height)), _bucket(4 * _MAX_PAIRS.get(key, 0)) // 4)
        if cap > 0:
            point_list = torch.empty((cap,), dtype=torch.int32, device=dev)
    ticket, sorted_ = ctypes.c_uint32(0), ctypes.c_int(0)
    if _zhx_on(cuda_args):
        # the native timer log wants the reference's two stages apart (analyze_statistic.py:1972-1991): two host calls
        with zhx_range(cuda_args, "24 updateDistributedStatLocally.updateTileTouched time"):
            check(lib.gsr_bin_prepare_async(P, width, height, _ptr(means2D), _ptr(depths), _ptr(radii),
                                            _ptr(conic_opacity), _ptr(compute_locally), _ptr(prep), prep_bytes,
                                            ctypes.byref(ticket), stream), "gsr_bin_prepare_async")
        if cap > 0 and ticket.value != 0:
            with zhx_range(cuda_args, "50 SortPairs time"):
                check(lib.gsr_bin_sort_bounded(P, width, height, _ptr(compute_locally), _ptr(prep), cap, _ptr(kept),
                                               kept.numel(), _ptr(point_list), _ptr(ranges), stream),
                      "gsr_bin_sort_bounded")
            sorted_.value = 1
    else:
        # ONE host-side call: K3-K4 and the tile sort at the scratch's capacity (include/gsraster.h)
        check(lib.gsr_bin_speculative_async(P, width, height, _ptr(means2D), _ptr(depths), _ptr(radii),
                                            _ptr(conic_opacity), _ptr(compute_locally), _ptr(prep), prep_bytes,
                                            cap if cap > 0 else 0, _ptr(kept) if cap > 0 else None,
                                            kept.numel() if cap > 0 else 0, _ptr(point_list) if cap > 0 else None,
                                            _ptr(ranges), ctypes.byref(ticket), ctypes.byref(sorted_), stream),
              "gsr_bin_speculative_async")
    pend = PendingPairs(ticket.value, bool(sorted_.value), cap, P, width, height, compute_locally, prep, ranges, dev,
                        current_stream(dev), key, cuda_args)
    if kernel_timer.enabled and ticket.value:
        # bench.py's instrumented replay only (a device read-back): the row segments R of this view, for the bytes of the
        # row-major pipeline (include/gsraster.h: gsr_bin_segments_offset)
        off = int(lib.gsr_bin_segments_offset(P, width, height))
        _LAST_SEGMENTS[0] = int(prep[off:off + 4].view(torch.int32).item())
    if defer and sorted_.value:
        return point_list, ranges, pend
    D, again = pend.finish()
    if again is not None:
        return again, ranges, D
    return point_list[:max(D, 1)], ranges, D


class _RenderGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2D, conic_opacity, rgb, depths, radii, compute_locally, raster_settings, cuda_args,
                token=None):
        # `token`: an optional 1-element tensor that only ties this node into the caller's autograd graph (the mirror
        # chains its per-camera exchanges through it so that every rank runs their backward collectives in one order)
        ctx.set_materialize_grads(False)
        rs = raster_settings
        means2D, conic_opacity, rgb = _f32c(means2D, "means2D"), _f32c(conic_opacity, "conic_opacity"), _f32c(rgb, "rgb")
        depths = _f32c(depths, "depths")
        if not radii.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization: `radii` must live on the gfx950 device")
        radii = radii.to(torch.int32).contiguous()
        H, W = int(rs.image_height), int(rs.image_width)
        gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
        P = means2D.shape[0]
        dev = means2D.device
        if compute_locally is None:
            mask = torch.ones((gy * gx,), dtype=torch.uint8, device=dev)
        else:
            if compute_locally.numel() != gx * gy:
                raise ValueError(f"compute_locally must have TILE_Y*TILE_X = {gy}x{gx} entries")
            mask = compute_locally.to(device=dev).contiguous().view(-1)
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        bg = _f32c(rs.bg, "bg")
        # per call (the mirror asks through cuda_args, so a module-wide mode a user chose is never overridden), else
        # the module-wide mode
        timing = (cuda_args.get("_gsr_timing") if isinstance(cuda_args, dict) else None) or _timing_mode()
        if _CAPTURE[0] is not None:
            timing = "off"  # events recorded inside a capture cannot be read
        stats = cuda_args.get("stats_collector") if isinstance(cuda_args, dict) else None
        with _on(dev):
            if timing != "off":
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record(current_stream(dev))
            if _CAPTURE[0] is not None:
                _CAPTURE[0].stamp("fwd0", id(stats))
            with kernel_timer.range("binning", P=P, tiles=gx * gy) as kt:
                # (round 6) the pair count is looked at AFTER K8 has been launched -- by the caller, after ALL its cameras
                # have been launched, when it hands a list through cuda_args["_gsr_pending"] (gaussian_renderer.render_final)
                point_list, ranges, D = bin_gaussians(means2D, depths, radii, conic_opacity, mask, W, H, cuda_args,
                                                      defer=True)
            pend = D if isinstance(D, PendingPairs) else None
            kt.meta["D"] = D = (0 if pend is not None else D)
            if kernel_timer.enabled:
                kt.meta["R"] = _LAST_SEGMENTS[0]
            out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            final_T = torch.empty((H, W), dtype=torch.float32, device=dev)
            n_contrib = torch.empty((H, W), dtype=torch.int32, device=dev)
            # bench bookkeeping: the launch's local pixel count is derived from the mask AFTER the run (no sync here)
            ctx.px_meta = dict(mask=mask, W=W, H=H) if kernel_timer.enabled else {}
            ctx.cuda_args = cuda_args
            # the tile rows of the caller's band, when it names them (the mirror does: cuda_args["_gsr_band"]; the
            # mask must be false outside): the launches then cover the band's tiles only
            band = cuda_args.get("_gsr_band") if (isinstance(cuda_args, dict) and _BAND_GRID[0]) else None
            row_lo, row_hi = (int(band[0]), int(band[1])) if band else (0, 0)
            # (-1, capacity): the band is device data -- the mask's row hull, left behind the range table by the tile
            # sort -- and the launches are sized for `capacity` tile rows (include/gsraster.h; graphed_step.py)
            band_rows = row_hi if row_lo == -1 else row_hi - row_lo
            thin = 0 < band_rows * gx <= 2048  # at most two rounds of resident workgroups
            # list segments for the backward (include/gsraster.h: gsr_render_forward_seg) when a backward can follow and
            # the band is THIN: measured (profiles/r04_ab_segments.txt) -9 us net on a 1/8 band, nothing on a whole image
            # (its lists finish in staggered rounds anyway) where the checkpoints only cost the forward 4-9 us
            seg_ws, seg_bytes = None, 0
            if any(ctx.needs_input_grad[:3]) and (_SEGMENTS[0] == "always" or (_SEGMENTS[0] and thin)):
                seg_bytes = int(lib.gsr_render_seg_bytes(W, H))
                seg_ws = torch.empty((seg_bytes,), dtype=torch.uint8, device=dev)
            lists = [point_list]  # (a list: a late pair count may replace the point_list, see settle below)
            # K10's [P,9] gradient record (means2D 0:2, rgb 2:5, conic_opacity 5:9) is allocated HERE when a backward can
            # follow, and cleared by K8's own workgroups (include/gsraster.h: gsr_render_forward_seg_z) instead of by a
            # 36 MB fill launch at the head of the backward
            record = None
            if any(ctx.needs_input_grad[:3]) and P > 0 and _ZERO_IN_FORWARD[0]:
                record = torch.empty((P, 9), dtype=torch.float32, device=dev)
            ctx.record = record

            def launch_k8(meta_D):
                with kernel_timer.range("composite_forward", P=P, D=meta_D, **ctx.px_meta) as k8t, \
                        zhx_range(cuda_args, "70 render time"):
                    check(lib.gsr_render_forward_seg_z(P, W, H, _ptr(ranges), _ptr(lists[0]), _ptr(means2D),
                                                       _ptr(conic_opacity), _ptr(rgb), _ptr(mask), _ptr(bg), _ptr(out),
                                                       _ptr(final_T), _ptr(n_contrib), _ptr(seg_ws), seg_bytes, row_lo,
                                                       row_hi, _ptr(record), 36 * P if record is not None else 0,
                                                       _stream()), "gsr_render_forward_seg_z")
                return k8t

            k8t = launch_k8(D)
            if _CAPTURE[0] is not None:
                _CAPTURE[0].stamp("fwd1", id(stats))
            ctx.seg = (seg_ws, seg_bytes, row_lo, row_hi)
            ctx.lists = lists
            ctx.num_rendered = D
            _RenderGaussians.last_num_rendered = D
            if pend is not None:
                def settle(ctx=ctx, pend=pend, kt=kt, k8t=k8t, lists=lists):
                    D, again = pend.finish()
                    kt.meta["D"] = k8t.meta["D"] = D
                    ctx.num_rendered = D
                    _RenderGaussians.last_num_rendered = D
                    if again is not None:  # the capacity did not hold: K8 drew the background; draw the view now
                        lists[0] = again
                        with _on(pend.dev), torch.cuda.stream(pend.stream):
                            launch_k8(D)
                    return D

                collector = cuda_args.get("_gsr_pending") if isinstance(cuda_args, dict) else None
                if isinstance(collector, list):  # (the caller settles after its last camera's launches)
                    collector.append(settle)
                else:
                    settle()
            if timing != "off":
                ev1.record(current_stream(dev))
                ctx.fwd_events = (ev0, ev1)
        if stats is not None:
            # placeholders are real floats so that a forward-only caller can read them; the backward
            # replaces them with measured values (workload_division.py:953-957 reads them afterwards)
            stats.setdefault("forward_render_time", 0.0)
            stats.setdefault("backward_render_time", 0.0)
        ctx.raster_settings = rs
        ctx.cuda_args = cuda_args
        ctx.timing = timing
        # the image itself is an input of the segmented backward (the colour behind a boundary is reconstructed from it):
        # saved through autograd, so that an in-place edit of the output raises instead of corrupting gradients, and
        # no ctx -> output -> grad_fn cycle keeps the buffers alive until the cyclic collector runs
        ctx.save_for_backward(means2D, conic_opacity, rgb, mask, bg, ranges, final_T, n_contrib,
                              out if seg_ws is not None else None)
        ctx.mark_non_differentiable(n_contrib)
        return out, n_contrib

    @staticmethod
    def backward(ctx, g_out, _g_ncontrib):
        global _last_backward_ms
        rs = ctx.raster_settings
        means2D, conic_opacity, rgb, mask, bg, ranges, final_T, n_contrib, out_img = ctx.saved_tensors
        point_list = ctx.lists[0]  # (not a saved tensor: a late pair count may have replaced it, see forward)
        H, W = int(rs.image_height), int(rs.image_width)
        P = means2D.shape[0]
        dev = means2D.device
        if g_out is None:
            return None, None, None, None, None, None, None, None, None
        g_out = g_out.float().contiguous()
        # ONE [P,9] record (means2D 0:2, rgb 2:5, conic_opacity 5:9): K10 flushes a (tile, Gaussian) pair's nine sums
        # from nine adjacent lanes into one row; K11 and the exchange read the record through its row stride
        # (cleared by the forward's composite kernel when it was allocated there; a second backward over the same
        # forward -- retain_graph, gradcheck -- takes a fresh one and the fill)
        record, ctx.record = ctx.record, None
        record_is_zero = record is not None
        if record is None:
            record = torch.empty((P, 9), dtype=torch.float32, device=dev)
        d_means2D, d_rgb, d_conic_opacity = record[:, 0:2], record[:, 2:5], record[:, 5:9]
        timing = ctx.timing
        with _on(dev):
            if timing != "off":
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record(current_stream(dev))
            cap_stats = ctx.cuda_args.get("stats_collector") if isinstance(ctx.cuda_args, dict) else None
            if _CAPTURE[0] is not None:
                _CAPTURE[0].stamp("bwd0", id(cap_stats))
            with kernel_timer.range("composite_backward", P=P, D=ctx.num_rendered, **ctx.px_meta), \
                    zhx_range(ctx.cuda_args, "b10 render time"):
                seg_ws, seg_bytes, row_lo, row_hi = ctx.seg
                check(lib.gsr_render_backward_seg_z(P, W, H, _ptr(ranges), _ptr(point_list), _ptr(means2D),
                                                    _ptr(conic_opacity), _ptr(rgb), _ptr(mask), _ptr(bg), _ptr(final_T),
                                                    _ptr(n_contrib), _ptr(g_out), _ptr(record), _ptr(out_img),
                                                    _ptr(seg_ws), seg_bytes, row_lo, row_hi, 1 if record_is_zero else 0,
                                                    _stream()), "gsr_render_backward_seg_z")
            if _CAPTURE[0] is not None:
                _CAPTURE[0].stamp("bwd1", id(cap_stats))
            if timing != "off":
                ev1.record(current_stream(dev))
        stats = ctx.cuda_args.get("stats_collector") if isinstance(ctx.cuda_args, dict) else None
        if stats is not None and timing != "off":
            f0, f1 = ctx.fwd_events
            if timing == "sync":
                ev1.synchronize()
                _last_backward_ms = float(ev0.elapsed_time(ev1))
                stats["forward_render_time"] = float(f0.elapsed_time(f1))
                stats["backward_render_time"] = _last_backward_ms
            else:  # "stale" / "deferred": never wait for the device here
                if timing == "deferred":
                    stats["_fwd_events"] = (f0, f1)
                    stats["_bwd_events"] = (ev0, ev1)
                if f1.query():
                    stats["forward_render_time"] = float(f0.elapsed_time(f1))
                pend = getattr(_RenderGaussians, "_pending", None)
                if pend is not None and pend[1].query():
                    _last_backward_ms = float(pend[0].elapsed_time(pend[1]))
                _RenderGaussians._pending = (ev0, ev1)
                stats["backward_render_time"] = float(_last_backward_ms)
        return d_means2D, d_conic_opacity, d_rgb, None, None, None, None, None, None


class GaussianRasterizer(nn.Module):
    """one instance per camera per iteration (gaussian_renderer/__init__.py:945)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def preprocess_gaussians(self, means3D, scales, rotations, shs, opacities, cuda_args=None):
        """-> (means2D [N,2], rgb [N,3], conic_opacity [N,4], radii int32 [N], depths [N]);
        radii == 0 <=> culled.  means2D supports .retain_grad(); its gradient is in NDC-scaled units
        (pixel gradient x (W/2, H/2)), the convention densification thresholds against
        (scene/gaussian_model.py:1046-1064)."""
        return _PreprocessGaussians.apply(means3D, scales, rotations, shs, opacities, self.raster_settings, cuda_args)

    def preprocess_gaussians_raw(self, xyz, scaling, rotation, features_dc, features_rest, opacity, cuda_args=None):
        """same outputs as preprocess_gaussians, from GaussianModel's RAW parameters (_xyz, _scaling, _rotation,
        _features_dc, _features_rest, _opacity): the getters' exp / normalize / sigmoid / cat are fused into the
        kernels.  Extension of this build (the reference's op takes activated tensors)."""
        return _PreprocessGaussiansRaw.apply(xyz, scaling, rotation, features_dc, features_rest, opacity,
                                             self.raster_settings, cuda_args)

    def render_gaussians(self, means2D, conic_opacity, rgb, depths, radii, compute_locally,
                         extended_compute_locally=None, cuda_args=None):
        """-> (image [3,H,W], n_render, n_consider, n_contrib).  Pixels of tiles with
        compute_locally == False are exactly 0.  The last three values are debug statistics the
        reference's callers discard (gaussian_renderer/__init__.py:1271): n_render = number of
        (tile, Gaussian) pairs as a Python int (0 when the caller collects the late pair counts itself through
        cuda_args["_gsr_pending"]: the count is then known when it settles them), n_consider = None, n_contrib =
        per-pixel int32 map."""
        if cuda_args is None:
            cuda_args = {}
        fn = _RenderGaussians
        token = cuda_args.get("_exchange_token") if isinstance(cuda_args, dict) else None
        image, n_contrib = fn.apply(means2D, conic_opacity, rgb, depths, radii, compute_locally,
                                    self.raster_settings, cuda_args, token)
        return image, getattr(fn, "last_num_rendered", None), None, n_contrib


# ------------------------------------------------------------------------------- N1: fused loss
class _FusedL1SSIMBand(torch.autograd.Function):
    """(sum |band - gt|, sum ssim_map(band, gt)) of rows [y0, y1) of `image` [C,H,W] against the uint8
    ground-truth band [C, y1-y0, W]; the two sums are what final_system_loss_computation divides by
    H*W*3 (gaussian_renderer/loss_distribution.py:2567-2576)."""

    @staticmethod
    def forward(ctx, image, gt_u8, y0, y1):
        if not image.is_cuda:
            raise RuntimeError("fused_l1_ssim_band: device tensors required (no CPU fallback)")
        image = image.float().contiguous()
        C, H, W = image.shape
        rows = y1 - y0
        gt_u8 = gt_u8.contiguous()
        if gt_u8.dtype != torch.uint8 or tuple(gt_u8.shape) != (C, rows, W):
            raise ValueError(f"gt band must be uint8 [{C},{rows},{W}], got {gt_u8.dtype} {tuple(gt_u8.shape)}")
        dev = image.device
        need_grad = ctx.needs_input_grad[0]
        nb = lib.gsr_l1_ssim_num_partials(C, rows, W)
        partials = torch.empty((max(nb, 1), 2), dtype=torch.float32, device=dev)
        maps = torch.empty((3, C, rows, W), dtype=torch.float32, device=dev) if need_grad else None
        band_ptr = ctypes.c_void_p(image.data_ptr() + 4 * y0 * W)
        with _on(dev), kernel_timer.range("l1_ssim_forward", Px=rows * W):
            check(lib.gsr_l1_ssim_forward(C, rows, W, band_ptr, H * W, _ptr(gt_u8), _ptr(partials),
                                          _ptr(maps[0]) if need_grad else None, _ptr(maps[1]) if need_grad else None,
                                          _ptr(maps[2]) if need_grad else None, _stream()), "gsr_l1_ssim_forward")
        sums = partials[:nb].sum(dim=0)
        ctx.y0, ctx.y1 = y0, y1
        if need_grad:
            ctx.save_for_backward(image, gt_u8, maps)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        image, gt_u8, maps = ctx.saved_tensors
        C, H, W = image.shape
        y0, y1 = ctx.y0, ctx.y1
        rows = y1 - y0
        dev = image.device
        g_l1 = (torch.zeros((), device=dev) if g_l1 is None else g_l1).float().contiguous()
        g_ssim = (torch.zeros((), device=dev) if g_ssim is None else g_ssim).float().contiguous()
        grad = torch.empty_like(image) if rows == H else torch.zeros_like(image)
        band_ptr = ctypes.c_void_p(image.data_ptr() + 4 * y0 * W)
        gband_ptr = ctypes.c_void_p(grad.data_ptr() + 4 * y0 * W)
        with _on(dev), kernel_timer.range("l1_ssim_backward", Px=rows * W):
            check(lib.gsr_l1_ssim_backward(C, rows, W, band_ptr, H * W, _ptr(gt_u8), _ptr(maps[0]), _ptr(maps[1]),
                                           _ptr(maps[2]), _ptr(g_l1), _ptr(g_ssim), 1.0, 1.0, gband_ptr, H * W,
                                           _stream()),
                  "gsr_l1_ssim_backward")
        return grad, None, None, None


def fused_l1_ssim_band(image, gt_u8, y0, y1):
    """-> (sum of |x - gt/255| , sum of the SSIM map) over rows [y0, y1) of image [C,H,W]"""
    return _FusedL1SSIMBand.apply(image, gt_u8, int(y0), int(y1))


class _FusedBandLoss(torch.autograd.Function):
    """loss = (1 - lambda) * Ll1 + lambda * (1 - ssim) of one rendered row band, Ll1 = sum|x - gt| / n and
    ssim = sum(ssim_map) / n with n = H*W*3 of the FULL image (gaussian_renderer/loss_distribution.py:2567-2576,
    2627-2629), as two launches forward (map kernel + finalize) and two backward (scalar scale + map kernel) --
    the reference's ~30 elementwise launches around its conv2d-based maps are gone.  Returns
    (loss, Ll1, ssim); the last two are detached (logging only)."""

    _coef_cache = {}

    @staticmethod
    def forward(ctx, image, gt_u8, y0, y1, lambda_dssim, n, band_rows=None):
        if not image.is_cuda:
            raise RuntimeError("fused_band_loss: device tensors required (no CPU fallback)")
        ctx.set_materialize_grads(False)
        image = image.float().contiguous()
        C, H, W = image.shape
        gt_u8 = gt_u8.contiguous()
        # band_rows (int32 [2] on the device: { y0, y1 }): the band is DEVICE data, `gt_u8` [C, capacity, W] carries it in
        # the first y1 - y0 rows of every channel and y0 / y1 are not looked at (graphed_step.py: one captured launch for
        # every band; include/gsraster.h: gsr_l1_ssim_forward_band)
        dyn = band_rows is not None
        if dyn and (band_rows.dtype != torch.int32 or not band_rows.is_cuda or band_rows.numel() < 2):
            raise ValueError("band_rows must be an int32 device tensor { y0, y1 }")
        rows = int(gt_u8.shape[1]) if dyn else y1 - y0
        if gt_u8.dtype != torch.uint8 or tuple(gt_u8.shape) != (C, rows, W):
            raise ValueError(f"gt band must be uint8 [{C},{rows},{W}], got {gt_u8.dtype} {tuple(gt_u8.shape)}")
        dev = image.device
        need_grad = ctx.needs_input_grad[0]
        nb = lib.gsr_l1_ssim_num_partials(C, rows, W)
        partials = torch.empty((max(nb, 1), 2), dtype=torch.float32, device=dev)
        maps = torch.empty((3, C, rows, W), dtype=torch.float32, device=dev) if need_grad else None
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        c_l1, c_ssim = (1.0 - lambda_dssim) / n, -lambda_dssim / n
        m = [_ptr(maps[i]) if need_grad else None for i in range(3)]
        with _on(dev):
            with kernel_timer.range("l1_ssim_forward", Px=rows * W):
                if dyn:
                    check(lib.gsr_l1_ssim_forward_band(C, rows, W, _ptr(image), H * W, _ptr(gt_u8), _ptr(partials), m[0],
                                                       m[1], m[2], _ptr(band_rows), _stream()),
                          "gsr_l1_ssim_forward_band")
                else:
                    band_ptr = ctypes.c_void_p(image.data_ptr() + 4 * y0 * W)
                    check(lib.gsr_l1_ssim_forward(C, rows, W, band_ptr, H * W, _ptr(gt_u8), _ptr(partials), m[0], m[1],
                                                  m[2], _stream()), "gsr_l1_ssim_forward")
            check(lib.gsr_l1_ssim_finalize(nb, _ptr(partials), c_l1, c_ssim, lambda_dssim, 1.0 / n, _ptr(out3),
                                           _stream()), "gsr_l1_ssim_finalize")
        ctx.y0, ctx.y1, ctx.coef, ctx.band_rows = y0, y1, (c_l1, c_ssim), band_rows
        if need_grad:
            ctx.save_for_backward(image, gt_u8, maps)
        loss, Ll1, ssim = out3[0], out3[1], out3[2]
        ctx.mark_non_differentiable(Ll1, ssim)
        return loss, Ll1, ssim

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None:
            return None, None, None, None, None, None, None
        image, gt_u8, maps = ctx.saved_tensors
        C, H, W = image.shape
        y0, y1 = ctx.y0, ctx.y1
        dyn = ctx.band_rows is not None
        rows = int(gt_u8.shape[1]) if dyn else y1 - y0
        dev = image.device
        g_loss = g_loss if (g_loss.dtype == torch.float32 and g_loss.is_contiguous()) else g_loss.float().contiguous()
        # (dL/dS_l1, dL/dS_ssim) = dL/dloss * (c_l1, c_ssim): the product is formed inside the kernel
        grad = torch.empty_like(image) if (rows == H and not dyn) else torch.zeros_like(image)
        with _on(dev), kernel_timer.range("l1_ssim_backward", Px=rows * W):
            if dyn:
                check(lib.gsr_l1_ssim_backward_band(C, rows, W, _ptr(image), H * W, _ptr(gt_u8), _ptr(maps[0]),
                                                    _ptr(maps[1]), _ptr(maps[2]), _ptr(g_loss), _ptr(g_loss),
                                                    float(ctx.coef[0]), float(ctx.coef[1]), _ptr(grad), H * W,
                                                    _ptr(ctx.band_rows), _stream()), "gsr_l1_ssim_backward_band")
            else:
                band_ptr = ctypes.c_void_p(image.data_ptr() + 4 * y0 * W)
                gband_ptr = ctypes.c_void_p(grad.data_ptr() + 4 * y0 * W)
                check(lib.gsr_l1_ssim_backward(C, rows, W, band_ptr, H * W, _ptr(gt_u8), _ptr(maps[0]), _ptr(maps[1]),
                                               _ptr(maps[2]), _ptr(g_loss), _ptr(g_loss), float(ctx.coef[0]),
                                               float(ctx.coef[1]), gband_ptr, H * W, _stream()),
                      "gsr_l1_ssim_backward")
        return grad, None, None, None, None, None, None


def fused_band_loss(image, gt_u8, y0, y1, lambda_dssim, n, band_rows=None):
    """-> (loss, Ll1, ssim) of rows [y0, y1) of image [C,H,W]; n = H*W*3 of the full image.  band_rows (int32 [2] on
    the device): the rows are device data and gt_u8 is a [C, capacity, W] buffer (see _FusedBandLoss.forward)"""
    return _FusedBandLoss.apply(image, gt_u8, int(y0), int(y1), float(lambda_dssim), float(n), band_rows)


# ------------------------------------------------------------------------- a19: fused activations
class _FusedActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, opacity, features_dc, features_rest):
        ts = [scaling, rotation, opacity, features_dc, features_rest]
        if not all(t.is_cuda for t in ts):
            raise RuntimeError("fused_activations: device tensors required (no CPU fallback)")
        scaling, rotation, opacity, features_dc, features_rest = [t.float().contiguous() for t in ts]
        N, rest = scaling.shape[0], features_rest.shape[1]
        dev = scaling.device
        scales = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rotations = torch.empty((N, 4), dtype=torch.float32, device=dev)
        opacities = torch.empty((N, 1), dtype=torch.float32, device=dev)
        shs = torch.empty((N, 1 + rest, 3), dtype=torch.float32, device=dev)
        with _on(dev), kernel_timer.range("activate_forward"):
            check(lib.gsr_activate_forward(N, rest, _ptr(scaling), _ptr(rotation), _ptr(opacity), _ptr(features_dc),
                                           _ptr(features_rest), _ptr(scales), _ptr(rotations), _ptr(opacities),
                                           _ptr(shs), _stream()), "gsr_activate_forward")
        ctx.save_for_backward(rotation, scales, opacities)
        ctx.rest = rest
        return scales, rotations, opacities, shs

    @staticmethod
    def backward(ctx, g_scales, g_rotations, g_opacities, g_shs):
        rotation, scales, opacities = ctx.saved_tensors
        N, rest = scales.shape[0], ctx.rest
        dev = scales.device

        def z(g, shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.float().contiguous()

        g_scales, g_rotations = z(g_scales, (N, 3)), z(g_rotations, (N, 4))
        g_opacities, g_shs = z(g_opacities, (N, 1)), z(g_shs, (N, 1 + rest, 3))
        d_scaling = torch.empty((N, 3), dtype=torch.float32, device=dev)
        d_rotation = torch.empty((N, 4), dtype=torch.float32, device=dev)
        d_opacity = torch.empty((N, 1), dtype=torch.float32, device=dev)
        d_dc = torch.empty((N, 1, 3), dtype=torch.float32, device=dev)
        d_rest = torch.empty((N, rest, 3), dtype=torch.float32, device=dev)
        with _on(dev), kernel_timer.range("activate_backward"):
            check(lib.gsr_activate_backward(N, rest, _ptr(rotation), _ptr(scales), _ptr(opacities), _ptr(g_scales),
                                            _ptr(g_rotations), _ptr(g_opacities), _ptr(g_shs), _ptr(d_scaling),
                                            _ptr(d_rotation), _ptr(d_opacity), _ptr(d_dc), _ptr(d_rest), _stream()),
                  "gsr_activate_backward")
        return d_scaling, d_rotation, d_opacity, d_dc, d_rest


def fused_activations(scaling, rotation, opacity, features_dc, features_rest):
    """(scales, rotations, opacities, shs) = (exp, normalize, sigmoid, cat) of the raw parameters in one
    kernel each way -- GaussianModel.get_scaling/get_rotation/get_opacity/get_features"""
    return _FusedActivations.apply(scaling, rotation, opacity, features_dc, features_rest)


def exchange_need(means2D_all, radii_all, bands, width, height):
    """K2 for the whole camera batch in the exchange's layout: means2D_all [B,P,2], radii_all int32 [B,P],
    bands int32 [B,W,2] (tile rows [lo,hi) of camera k rendered by global rank g) ->
    (need bool [W,B,P], counts int32 [W,B])."""
    m2 = _f32c(means2D_all.detach(), "means2D_all")
    if radii_all.dtype != torch.int32 or not radii_all.is_contiguous() or bands.dtype != torch.int32:
        raise ValueError("radii_all / bands must be contiguous int32")
    B, P = radii_all.shape
    W = bands.shape[1]
    bands = bands.to(m2.device).contiguous()
    need = torch.empty((W, B, P), dtype=torch.bool, device=m2.device)
    counts = torch.empty((W, B), dtype=torch.int32, device=m2.device)
    check(lib.gsr_exchange_need(P, B, W, width, height, _ptr(m2), _ptr(radii_all), _ptr(bands), _ptr(need),
                                _ptr(counts), _stream()), "gsr_exchange_need")
    return need, counts


def exchange_count(means2D_all, radii_all, bands, k0, nb, width, height):
    """K2 counted, not materialised: for the cameras [k0, k0 + nb) of the camera-major state (means2D_all fp32 [B,P,2],
    radii_all int32 [B,P]; bands int32 [B,W,2] on the device) -> (chunkcnt int32 [W * nb, chunks], counts int32 [W, nb])"""
    m2 = _f32c(means2D_all.detach(), "means2D_all")
    if radii_all.dtype != torch.int32 or not radii_all.is_contiguous() or bands.dtype != torch.int32 or \
            not bands.is_cuda or not bands.is_contiguous():
        raise ValueError("radii_all / bands must be contiguous int32 device tensors")
    B, P = radii_all.shape
    W = bands.shape[1]
    nchunk = lib.gsr_exchange_chunks(P)
    chunkcnt = torch.empty((W * nb, max(nchunk, 1)), dtype=torch.int32, device=m2.device)
    counts = torch.empty((W, nb), dtype=torch.int32, device=m2.device)
    with _on(m2.device):
        check(lib.gsr_exchange_count(P, B, k0, nb, W, width, height, _ptr(m2), _ptr(radii_all), _ptr(bands),
                                     _ptr(chunkcnt), _ptr(counts), _stream()), "gsr_exchange_count")
    return chunkcnt, counts


def exchange_pack(means2D_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt, segment_offsets, n_send, k0, nb,
                  width, height, count_cameras=None, count_first=None):
    """-> (msg fp32 [n_send, 11], send_idx int32 [n_send]): the records of the cameras [k0, k0 + nb) in
    (destination, camera, local index) order; segment_offsets: python ints [W * nb] (first row of each segment);
    chunkcnt: what exchange_count returned for the camera range [count_first, count_first + count_cameras) (default:
    the same range)"""
    B, P = radii_all.shape
    W = bands.shape[1]
    dev = radii_all.device
    msg = torch.empty((n_send, 11), dtype=torch.float32, device=dev)
    send_idx = torch.empty((n_send,), dtype=torch.int32, device=dev)
    if len(segment_offsets) != W * nb:
        raise ValueError("segment_offsets must have W * nb entries")
    seg = (ctypes.c_int32 * (W * nb))(*segment_offsets)
    with _on(dev):
        check(lib.gsr_exchange_pack(P, B, k0, nb, W, width, height, nb if count_cameras is None else count_cameras,
                                    k0 if count_first is None else count_first, _ptr(means2D_all), _ptr(rgb_all), _ptr(co_all),
                                    _ptr(radii_all), _ptr(depths_all), _ptr(bands), _ptr(chunkcnt), seg, n_send,
                                    _ptr(msg), _ptr(send_idx), _stream()), "gsr_exchange_pack")
    return msg, send_idx


def exchange_pack_slab(means2D_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt, counts, capacities, k0, nb,
                       width, height, count_cameras=None, count_first=None):
    """the exchange's pack WITHOUT this step's counts on the host (include/gsraster.h: gsr_exchange_pack_slab):
    capacities = python ints [W * nb], rows reserved per (destination, camera of [k0, k0 + nb)); `counts` = the DEVICE
    tensor exchange_count returned with `chunkcnt`.  -> (msg fp32 [sum(capacities), 11], send_idx int32
    [sum(capacities)]): every slab holds its records at the front (reference order), then all-zero padding rows
    (radius 0) whose send_idx is -1; records that do not fit are dropped (the caller detects that from the counts)."""
    B, P = radii_all.shape
    W = bands.shape[1]
    dev = radii_all.device
    if len(capacities) != W * nb:
        raise ValueError("capacities must have W * nb entries")
    n_rows = int(sum(capacities))
    msg = torch.empty((n_rows, 11), dtype=torch.float32, device=dev)
    send_idx = torch.empty((n_rows,), dtype=torch.int32, device=dev)
    caps = (ctypes.c_int32 * (W * nb))(*[int(c) for c in capacities])
    with _on(dev):
        check(lib.gsr_exchange_pack_slab(P, B, k0, nb, W, width, height, nb if count_cameras is None else count_cameras,
                                         k0 if count_first is None else count_first, _ptr(means2D_all), _ptr(rgb_all),
                                         _ptr(co_all), _ptr(radii_all), _ptr(depths_all), _ptr(bands), _ptr(chunkcnt),
                                         _ptr(counts), caps, n_rows, _ptr(msg), _ptr(send_idx), _stream()),
              "gsr_exchange_pack_slab")
    return msg, send_idx


def exchange_unpack(recv):
    """received message fp32 [n, 11] -> (means2D [n,2], rgb [n,3], conic_opacity [n,4], radii int32 [n], depths [n]) in
    one launch (include/gsraster.h: gsr_exchange_unpack)"""
    if not recv.is_cuda or recv.dtype != torch.float32 or recv.dim() != 2 or recv.shape[1] != 11 or \
            not recv.is_contiguous():
        raise ValueError("recv must be a contiguous fp32 [n, 11] device tensor")
    n, dev = recv.shape[0], recv.device
    outs = [torch.empty((n, w), dtype=torch.float32, device=dev) for w in (2, 3, 4)]
    radii = torch.empty((n,), dtype=torch.int32, device=dev)
    depths = torch.empty((n,), dtype=torch.float32, device=dev)
    with _on(dev):
        check(lib.gsr_exchange_unpack(n, _ptr(recv), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(radii),
                                      _ptr(depths), _stream()), "gsr_exchange_unpack")
    return outs[0], outs[1], outs[2], radii, depths


def exchange_check(all_counts, caps_dev, W, B, me, rendered_mask, few, flag, host_copy=None):
    """capacity check of a captured exchange (include/gsraster.h: gsr_exchange_check): raises FLAG_SLAB / FLAG_FEW in
    the device word `flag`"""
    if all_counts.dtype != torch.int32 or caps_dev.dtype != torch.int32 or all_counts.numel() != W * W * B or \
            caps_dev.numel() != W * W * B or not all_counts.is_contiguous() or not caps_dev.is_contiguous():
        raise ValueError("all_counts / caps_dev must be contiguous int32 [W*W*B]")
    with _on(all_counts.device):
        check(lib.gsr_exchange_check(_ptr(all_counts), _ptr(caps_dev), W, B, me, int(rendered_mask), int(few),
                                     _ptr(flag), FLAG_SLAB, FLAG_FEW,
                                     host_copy.data_ptr() if host_copy is not None else None, _stream()),
              "gsr_exchange_check")


def zeros_async(shape, dtype, device):
    """torch.zeros through hipMemsetAsync on the current stream (the fill kernel torch launches runs at ~1 TB/s)"""
    t = torch.empty(shape, dtype=dtype, device=device)
    with _on(device):
        check(lib.gsr_zero_async(_ptr(t), t.numel() * t.element_size(), _stream()), "gsr_zero_async")
    return t


def scatter_add_rows(idx, src, n_rows, dst=None):
    """-> dst fp32 [n_rows, 9] with dst[idx[r]] += src[r] (rows of 9 floats; duplicates in idx accumulate; rows with
    idx[r] < 0 are skipped); `dst`: an existing contiguous [n_rows, 9] view to add into (default: a fresh zero tensor)"""
    src = _f32c(src, "src")
    if src.dim() != 2 or src.shape[1] != 9 or idx.dtype != torch.int32 or not idx.is_contiguous():
        raise ValueError("src must be [n, 9] fp32 and idx contiguous int32")
    if dst is None:
        dst = torch.zeros((n_rows, 9), dtype=torch.float32, device=src.device)
    elif tuple(dst.shape) != (n_rows, 9) or dst.dtype != torch.float32 or not dst.is_contiguous():
        raise ValueError("dst must be a contiguous fp32 [n_rows, 9] tensor")
    with _on(src.device):
        check(lib.gsr_scatter_add_rows(src.shape[0], _ptr(idx), _ptr(src), _ptr(dst), _stream()), "gsr_scatter_add_rows")
    return dst


def densify_stats(radii, grad, max_radii2D, accum, denom):
    """the per-iteration densification statistics in one launch (include/gsraster.h: gsr_densify_stats); in place.
    radii int32 [P]; grad float32 [P, >=2] with contiguous rows (any row stride: a column view of K10's record is fine);
    max_radii2D [P], accum [P,1], denom [P,1] float32 dense"""
    P = radii.shape[0]
    for t, n in ((radii, "radii"), (grad, "grad"), (max_radii2D, "max_radii2D"), (accum, "accum"), (denom, "denom")):
        if not t.is_cuda:
            raise RuntimeError(f"diff_gaussian_rasterization: `{n}` must live on the gfx950 device (no CPU fallback)")
    if radii.dtype != torch.int32 or grad.dtype != torch.float32 or grad.dim() != 2 or grad.stride(1) != 1 or \
            any(t.dtype != torch.float32 or not t.is_contiguous() for t in (max_radii2D, accum, denom)):
        raise ValueError("densify_stats: int32 radii, float32 grad rows and dense float32 accumulators expected")
    with _on(radii.device):
        check(lib.gsr_densify_stats(P, _ptr(radii.contiguous()), _ptr(grad), grad.stride(0) if P > 1 else 2,
                                    _ptr(max_radii2D), _ptr(accum), _ptr(denom), _stream()), "gsr_densify_stats")


def knn_mean_dist2(points):
    """mean squared distance of every point to its 3 nearest other points -- what the reference obtains from
    `simple_knn._C.distCUDA2` at scene creation (scene/gaussian_model.py:163-166).  points: fp32 [P,3] on the
    device; returns fp32 [P]."""
    pts = _f32c(points.detach(), "points")
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be [P,3]")
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    nbytes = lib.gsr_knn_workspace_bytes(P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    check(lib.gsr_knn_mean_dist2(P, _ptr(pts), _ptr(out), _ptr(ws), nbytes, _stream()), "gsr_knn_mean_dist2")
    return out


def group_rows(dest, num_groups):
    """stable grouping of row indices by destination: returns (order int32 [N], counts list of num_groups+1
    python ints; the last entry counts the rows whose destination is outside [0, num_groups) = dropped).
    One host read-back (the counts size what follows)."""
    if not dest.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization: `dest` must live on the gfx950 device (no CPU fallback)")
    dest = dest.to(torch.int32).contiguous()
    N = dest.shape[0]
    order = torch.empty(N, dtype=torch.int32, device=dest.device)
    counts = torch.empty(num_groups + 1, dtype=torch.int64, device=dest.device)
    nbytes = lib.gsr_group_rows_bytes(N)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dest.device)
    check(lib.gsr_group_rows(N, num_groups, _ptr(dest), _ptr(order), _ptr(counts), _ptr(ws), nbytes, _stream()),
          "gsr_group_rows")
    return order, counts.cpu().tolist()


def scatter_rows(order, n_in, srcs, dsts, row0=0):
    """dst_k[order[row0 + r]] = src_k[r] for r < n_in, all tensors in ONE launch (the inverse of gather_rows)"""
    return gather_rows(order, n_in, srcs, dsts, row0, _scatter=True)


def gather_rows(order, n_out, srcs, dsts=None, row0=0, _scatter=False):
    """dst_k[r] = src_k[order[row0 + r]] (order None: identity) for r < n_out, all tensors in ONE launch.
    srcs / dsts: 4-byte-element tensors whose rows are contiguous (dim 0 may be strided, e.g. a column block of a
    record matrix); dsts=None allocates dense outputs shaped like the sources.  Returns the dsts."""
    if dsts is None:
        dsts = [torch.empty((n_out,) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device) for s in srcs]
    K = len(srcs)
    if K == 0 or n_out == 0:
        return dsts
    if K > 32:
        for i in range(0, K, 32):
            gather_rows(order, n_out, srcs[i:i + 32], dsts[i:i + 32], row0, _scatter)
        return dsts

    def row_layout(t, what):
        if not t.is_cuda:
            raise RuntimeError(f"diff_gaussian_rasterization: {what} must live on the gfx950 device (no CPU fallback)")
        if t.element_size() != 4:
            raise ValueError(f"{what}: 4-byte elements expected, got {t.dtype}")
        w = 1
        for d in t.shape[1:]:
            w *= d
        inner = t[0] if t.shape[0] > 0 else None
        if inner is not None and inner.numel() > 1 and not inner.is_contiguous():
            raise ValueError(f"{what}: rows must be contiguous")
        return w, (t.stride(0) if t.dim() > 0 and t.shape[0] > 1 else w)

    widths, ss, ds = [], [], []
    for s_, d_ in zip(srcs, dsts):
        w, st = row_layout(s_, "source")
        w2, dt = row_layout(d_, "destination")
        if w != w2 or (s_ if _scatter else d_).shape[0] < n_out:
            raise ValueError("source / destination row shapes differ")
        widths.append(w)
        ss.append(st)
        ds.append(dt)
    VP = ctypes.c_void_p * K
    order_ptr = ctypes.c_void_p(order.data_ptr() + 4 * row0) if order is not None else ctypes.c_void_p(0)
    if order is not None and (order.dtype != torch.int32 or not order.is_contiguous()):
        raise ValueError("order must be a contiguous int32 tensor")
    fn, what = (lib.gsr_scatter_rows, "gsr_scatter_rows") if _scatter else (lib.gsr_gather_rows, "gsr_gather_rows")
    check(fn(n_out, order_ptr, K, VP(*[s_.data_ptr() for s_ in srcs]), VP(*[d_.data_ptr() for d_ in dsts]),
             (ctypes.c_int32 * K)(*widths), (ctypes.c_int64 * K)(*ss), (ctypes.c_int64 * K)(*ds), _stream()), what)
    return dsts


# ------------------------------------------------------------------------------------------ _C
class _CNamespace:
    """stand-in for the reference extension's pybind module `diff_gaussian_rasterization._C`."""

    @staticmethod
    def get_block_XY():
        bx, by, bz = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        check(lib.gsr_get_block_xy(ctypes.byref(bx), ctypes.byref(by), ctypes.byref(bz)), "gsr_get_block_xy")
        return bx.value, by.value, bz.value

    @staticmethod
    def get_local2j_ids_bool(image_height, image_width, mp_rank, mp_world_size, means2D, radii,
                             dist_global_strategy, cuda_args=None):
        """bool [P, mp_world_size]: Gaussian i touches a tile of band j
        (gaussian_renderer/workload_division.py:727-738)."""
        means2D = _f32c(means2D.detach(), "means2D")
        radii = radii.to(torch.int32).contiguous()
        div = dist_global_strategy.to(device=means2D.device, dtype=torch.int32).contiguous()
        if div.numel() != mp_world_size + 1:
            raise ValueError("dist_global_strategy must have world_size + 1 entries")
        P = means2D.shape[0]
        out = torch.empty((P, mp_world_size), dtype=torch.uint8, device=means2D.device)
        with _on(means2D.device):
            check(lib.gsr_get_local2j_ids_bool(P, int(image_width), int(image_height), int(mp_world_size),
                                               _ptr(means2D), _ptr(radii), _ptr(div), _ptr(out), _stream()),
                  "gsr_get_local2j_ids_bool")
        return out.view(torch.bool)

    @staticmethod
    def _dead(name):
        raise NotImplementedError(
            f"diff_gaussian_rasterization._C.{name}: only reachable from the reference's legacy (non-`final`) "
            "distribution modes, which train.py never selects (SURVEY.md F4)")

    @staticmethod
    def get_local2j_ids_bool_adjust_mode6(*a, **k):
        _CNamespace._dead("get_local2j_ids_bool_adjust_mode6")

    @staticmethod
    def get_touched_locally(*a, **k):
        _CNamespace._dead("get_touched_locally")

    @staticmethod
    def get_pixels_compute_locally_and_in_rect(*a, **k):
        _CNamespace._dead("get_pixels_compute_locally_and_in_rect")


_C = _CNamespace()


def load_image_tiles_by_pos(*a, **k):
    _CNamespace._dead("load_image_tiles_by_pos")


def merge_image_tiles_by_pos(*a, **k):
    _CNamespace._dead("merge_image_tiles_by_pos")
