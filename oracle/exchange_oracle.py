"""TEST INFRASTRUCTURE ONLY -- plain-torch restatements of the device side of the fused exchange
(include/gsraster.h: gsr_exchange_count / gsr_exchange_pack / gsr_scatter_add_rows), used (a) by the CPU `gloo` tests
of the host logic in gaussian_renderer._batched_exchange_final, where there is no GPU to run the HIP kernels, and (b) by
the -m gpu test that checks the HIP kernels against them.  The partition rule is K2's
(gaussian_renderer/workload_division.py:721-744 of the reference): Gaussian i goes to rank g for camera k iff its
3-sigma tile rect meets the tile rows [lo, hi) rank g renders of camera k."""
import torch

from oracle import torch_oracle as O

XCHUNK = 1024


def _hits(means2D_all, radii_all, bands, k, width, height):
    """bool [W, P]: rank g needs Gaussian i of camera k"""
    m2, radii = means2D_all[k].detach(), radii_all[k]
    minx, miny, maxx, maxy = O.get_rect(m2, radii, width, height)
    ok = (radii > 0) & (maxx > minx) & (maxy > miny)
    W = bands.shape[1]
    out = torch.zeros((W, m2.shape[0]), dtype=torch.bool)
    for g in range(W):
        lo, hi = int(bands[k, g, 0]), int(bands[k, g, 1])
        if hi > lo:
            out[g] = ok & (torch.clamp(miny, min=lo) < torch.clamp(maxy, max=hi))
    return out


def exchange_count(means2D_all, radii_all, bands, k0, nb, width, height):
    B, P = radii_all.shape
    W = bands.shape[1]
    nchunk = max((P + XCHUNK - 1) // XCHUNK, 1)
    chunkcnt = torch.zeros((W * nb, nchunk), dtype=torch.int32)
    counts = torch.zeros((W, nb), dtype=torch.int32)
    for kk in range(nb):
        h = _hits(means2D_all, radii_all, bands.cpu(), k0 + kk, width, height)
        for g in range(W):
            counts[g, kk] = int(h[g].sum())
            pad = torch.zeros(nchunk * XCHUNK, dtype=torch.int32)
            pad[:P] = h[g].int()
            chunkcnt[g * nb + kk] = pad.view(nchunk, XCHUNK).sum(1)
    return chunkcnt, counts


def exchange_pack(means2D_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt, segment_offsets, n_send, k0, nb,
                  width, height, count_cameras=None, count_first=None):
    B, P = radii_all.shape
    W = bands.shape[1]
    dt = means2D_all.dtype
    msg = torch.zeros((n_send, 11), dtype=dt)
    send_idx = torch.zeros((n_send,), dtype=torch.int32)
    for kk in range(nb):
        k = k0 + kk
        h = _hits(means2D_all, radii_all, bands.cpu(), k, width, height)
        for g in range(W):
            ids = h[g].nonzero().squeeze(1)
            o = segment_offsets[g * nb + kk]
            n = ids.numel()
            rows = slice(o, o + n)
            msg[rows, 0:2] = means2D_all[k].detach()[ids]
            msg[rows, 2:5] = rgb_all[k].detach()[ids]
            msg[rows, 5:9] = co_all[k].detach()[ids]
            # the radius travels as the BITS of an int32 in a float lane; float64 test runs carry the value instead
            msg[rows, 9] = radii_all[k][ids].to(torch.int32).view(torch.float32) if dt == torch.float32 \
                else radii_all[k][ids].to(dt)
            msg[rows, 10] = depths_all[k].detach()[ids]
            send_idx[rows] = (kk * P + ids).to(torch.int32)
    return msg, send_idx


def exchange_pack_slab(means2D_all, rgb_all, co_all, radii_all, depths_all, bands, chunkcnt, counts, capacities, k0, nb,
                       width, height, count_cameras=None, count_first=None):
    """gsr_exchange_pack_slab: capacity slabs per (destination, camera); records at the front in the reference order,
    overflowing records dropped, all-zero padding (radius 0) with send_idx -1 behind them"""
    B, P = radii_all.shape
    W = bands.shape[1]
    dt = means2D_all.dtype
    n_rows = int(sum(capacities))
    msg = torch.zeros((n_rows, 11), dtype=dt)
    send_idx = torch.full((n_rows,), -1, dtype=torch.int32)
    o = 0
    for g in range(W):
        for kk in range(nb):
            k = k0 + kk
            cap = int(capacities[g * nb + kk])
            ids = _hits(means2D_all, radii_all, bands.cpu(), k, width, height)[g].nonzero().squeeze(1)[:cap]
            rows = slice(o, o + ids.numel())
            msg[rows, 0:2] = means2D_all[k].detach()[ids]
            msg[rows, 2:5] = rgb_all[k].detach()[ids]
            msg[rows, 5:9] = co_all[k].detach()[ids]
            msg[rows, 9] = radii_all[k][ids].to(torch.int32).view(torch.float32) if dt == torch.float32 \
                else radii_all[k][ids].to(dt)
            msg[rows, 10] = depths_all[k].detach()[ids]
            send_idx[rows] = (kk * P + ids).to(torch.int32)
            o += cap
    return msg, send_idx


def exchange_unpack(recv):
    """gsr_exchange_unpack: the five tensors of the render op out of the [n, 11] message"""
    radii = recv[:, 9].contiguous().view(torch.int32) if recv.dtype == torch.float32 else recv[:, 9].to(torch.int32)
    return (recv[:, 0:2].contiguous(), recv[:, 2:5].contiguous(), recv[:, 5:9].contiguous(), radii,
            recv[:, 10].contiguous())


def zeros_async(shape, dtype, device):
    return torch.zeros(shape, dtype=dtype, device=device)


def scatter_add_rows(idx, src, n_rows, dst=None):
    if dst is None:
        dst = torch.zeros((n_rows, 9), dtype=src.dtype)
    keep = idx >= 0  # slab padding rows carry -1
    dst.index_add_(0, idx[keep].long(), src[keep])
    return dst
