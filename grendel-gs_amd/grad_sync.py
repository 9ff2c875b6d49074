"""Gradient synchronisation for the REPLICATED-Gaussian storage mode (SURVEY.md "next" N2).

Mirror of scene/gaussian_model.py:1332-1439 of the reference (sync_gradients_densely / _sparsely /
_fused_densely; `_fused_sparsely` raises NotImplementedError there, :1438-1439) and of its dispatcher
`sync_gradients_for_replicated_3dgs_storage` (:364-394).  In the reference this mode is DEAD code: with more
than one rank the live path forces Gaussian sharding, every Gaussian has one owner and gradients return through
the backward of the sparse all-to-all -- there is no gradient all-reduce.  It is provided because
north_star names it, with the fused sparse variant the reference left unimplemented:

  fused_sparse: ONE all-reduce of the touched-row mask + ONE all-reduce of a compact [nnz, 59] buffer holding the
  six parameters' gradient rows (xyz 3, f_dc 3, f_rest 45, opacity 1, scaling 3, rotation 4) instead of six
  masked all-reduces.  A ring all-reduce over xGMI is bound by ONE link (~153 GB/s), not by the 7-link mesh, so
  shipping only the rows some rank touched (typically 10-30 % in large scenes) is what matters.
"""
import torch
import torch.distributed as dist

import diff_gaussian_rasterization as _dgr  # the row primitives of the HIP library (no CPU fallback)

PARAMS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _grads(gaussians):
    return [getattr(gaussians, n).grad.data for n in PARAMS]


def sync_gradients_densely(gaussians, group):
    with torch.no_grad():
        for g in _grads(gaussians):
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)


def sync_gradients_fused_densely(gaussians, group):
    with torch.no_grad():
        grads = _grads(gaussians)
        n = grads[0].shape[0]
        widths = [g.numel() // max(n, 1) for g in grads]
        flat = torch.cat([g.reshape(n, -1) for g in grads], dim=1).contiguous()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        for g, part in zip(grads, torch.split(flat, widths, dim=1)):
            g.copy_(part.reshape(g.shape))


def touched_rows_mask(gaussians, group):
    """rows with a non-zero xyz gradient on ANY rank (the reference keys sparsity on _xyz.grad, :1353-1361)"""
    local = (gaussians._xyz.grad.data != 0).any(dim=1).to(torch.int32)
    dist.all_reduce(local, op=dist.ReduceOp.MAX, group=group)
    return local.bool()


def sync_gradients_sparsely(gaussians, group):
    """the reference's variant: six all-reduces over the rows of the union mask"""
    with torch.no_grad():
        mask = touched_rows_mask(gaussians, group)
        for g in _grads(gaussians):
            part = g[mask].contiguous()
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
            g[mask] = part
    return mask


def sync_gradients_fused_sparsely(gaussians, group):
    """one mask all-reduce + one compact [nnz, 59] all-reduce for all six parameters.  The row selection is computed
    ONCE on the device (gsr_group_rows: touched rows first, stable), ONE launch packs the touched rows of the six
    gradient tensors into the column blocks of the compact buffer (gsr_gather_rows), ONE launch writes the reduced
    rows back (gsr_scatter_rows); the reference's per-tensor variant (scene/gaussian_model.py:1350-1391) runs a
    boolean-index gather, an all-reduce and a masked write per tensor."""
    with torch.no_grad():
        mask = touched_rows_mask(gaussians, group)
        grads = _grads(gaussians)
        n = grads[0].shape[0]
        widths = [g.numel() // max(n, 1) for g in grads]
        order, counts = _dgr.group_rows((~mask).to(torch.int32), 1)  # group 0 = touched rows; one host read-back:
        nnz = counts[0]                                              # the message size, the same on every rank
        compact = torch.empty((nnz, sum(widths)), dtype=torch.float32, device=grads[0].device)
        cols, c = [], 0
        for w in widths:
            cols.append(compact[:, c:c + w])
            c += w
        flat = [g.reshape(n, -1) for g in grads]
        _dgr.gather_rows(order, nnz, flat, cols)
        dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=group)
        _dgr.scatter_rows(order, nnz, cols, flat)
    return mask


_MODES = {"dense": sync_gradients_densely, "sparse": sync_gradients_sparsely,
          "fused_dense": sync_gradients_fused_densely, "fused_sparse": sync_gradients_fused_sparsely}


def sync_gradients_for_replicated_3dgs_storage(gaussians, group, sync_grad_mode="fused_sparse",
                                               gaussians_distribution=False):
    """dispatcher with the reference's flag semantics (arguments/__init__.py:156-157): only acts when the
    Gaussians are REPLICATED (gaussians_distribution False) and there is more than one rank"""
    if sync_grad_mode not in _MODES:
        raise AssertionError(f"sync_grad_mode {sync_grad_mode} not supported.")
    if not gaussians_distribution and group.size() > 1:
        return _MODES[sync_grad_mode](gaussians, group)
    return None
