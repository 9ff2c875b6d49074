"""N4: row primitives (gsr_group_rows / gsr_gather_rows) and the densification mirror (densification_ops.py)
against the boolean-indexing restatement of the reference's methods (oracle/densify_oracle.py) -- bit-exact:
rows are moved, not recomputed, and torch.normal draws the same samples on the same device."""
import copy

import pytest
import torch

import densification_ops as D
import diff_gaussian_rasterization as dgr
import synthetic_scene as S
from fused_optim import FusedAdam
from oracle import densify_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,groups", [(1, 1), (5, 3), (4097, 1), (100000, 8), (300001, 255)])
def test_group_rows_is_stable_partition(device, n, groups):
    g = torch.Generator().manual_seed(n)
    dest = torch.randint(-2, groups + 2, (n,), generator=g, dtype=torch.int32)
    order, counts = dgr.group_rows(dest.to(device), groups)
    ref_order, ref_counts = O.group_rows(dest, groups)
    assert counts == ref_counts
    assert torch.equal(order.cpu(), ref_order)


def test_gather_rows_dense_and_record_matrix(device):
    g = torch.Generator().manual_seed(0)
    n = 50000
    srcs = [torch.rand(n, 3, generator=g), torch.rand(n, 15, 3, generator=g), torch.rand(n, generator=g),
            torch.randint(0, 99, (n, 4), generator=g, dtype=torch.int32)]
    dest = torch.randint(-1, 2, (n,), generator=g, dtype=torch.int32)
    dsrcs = [s.to(device) for s in srcs]
    order, counts = dgr.group_rows(dest.to(device), 2)
    sel = order[:counts[0]].cpu().long()
    outs = dgr.gather_rows(order, counts[0], dsrcs)
    for o, s in zip(outs, srcs):
        assert torch.equal(o.cpu(), s[sel])
    # second group through row0; identity order; packing into / unpacking from one record matrix
    outs1 = dgr.gather_rows(order, counts[1], dsrcs, None, row0=counts[0])
    sel1 = order[counts[0]:counts[0] + counts[1]].cpu().long()
    assert torch.equal(outs1[1].cpu(), srcs[1][sel1])
    widths = [3, 45, 1, 4]
    rec = torch.zeros(counts[0], sum(widths), device=device)
    cols, c = [], 0
    for w in widths:
        cols.append(rec[:, c:c + w])
        c += w
    flat = [s.view(torch.float32).reshape(n, -1) if s.dtype != torch.float32 else s.reshape(n, -1) for s in dsrcs]
    dgr.gather_rows(order, counts[0], flat, cols)
    back = dgr.gather_rows(None, counts[0], cols)
    assert torch.equal(back[1].cpu().reshape(-1, 15, 3), srcs[1][sel])
    assert torch.equal(back[3].view(torch.int32).cpu(), srcs[3][sel])
    with pytest.raises(RuntimeError, match="no CPU"):
        dgr.gather_rows(None, 4, [srcs[0]])


def _model(device, n=30000, seed=0, world=1):
    torch.manual_seed(seed)
    m = S.SyntheticGaussianModel(n, 320, 240, seed=seed, device=device, scale_coef=0.02)
    m.optimizer = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    for _ in range(2):  # populate exp_avg / exp_avg_sq
        for p in m.parameters():
            p.grad = torch.randn_like(p)
        m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    m.percent_dense = 0.01
    m.xyz_gradient_accum = torch.rand(n, 1, device=device) * 0.001
    m.denom = torch.randint(0, 4, (n, 1), device=device).float()  # zeros -> NaN grads -> 0 (reference rule)
    m.max_radii2D = torch.zeros(n, device=device)
    m.sum_visible_count_in_one_batch = torch.rand(n, device=device)
    m.send_to_gpui_cnt = torch.randint(0, 9, (n, world), dtype=torch.int, device=device)
    return m


def _state(m):
    out = {}
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        out[g["name"]] = p.detach()
        st = m.optimizer.state[p]
        out[g["name"] + ".exp_avg"] = st["exp_avg"]
        out[g["name"] + ".exp_avg_sq"] = st["exp_avg_sq"]
        assert p is getattr(m, D._ATTR[g["name"]])
    for s in D._STATS:
        out[s] = getattr(m, s)
    return out


def test_densify_and_prune_matches_reference_restatement(device):
    a, b = _model(device), _model(device)
    extent = 4.0
    torch.manual_seed(123)
    D.densify_and_prune(a, 0.0002, 0.05, extent, 20)
    torch.manual_seed(123)
    O.densify_and_prune(b, 0.0002, 0.05, extent, 20)
    sa, sb = _state(a), _state(b)
    assert sa["xyz"].shape[0] != 30000  # something was cloned / split / pruned
    for k in sb:
        assert sa[k].shape == sb[k].shape, k
        assert torch.equal(sa[k], sb[k]), k
    # the optimizer keeps stepping on the new tensors
    for p in a.parameters():
        p.grad = torch.ones_like(p)
    a.optimizer.step()


def test_prune_points_all_and_none(device):
    m = _model(device, n=2000)
    ref = copy.deepcopy(_state(m))
    D.prune_points(m, torch.zeros(2000, dtype=torch.bool, device=device))
    for k, v in _state(m).items():
        assert torch.equal(v, ref[k]), k
    D.prune_points(m, torch.ones(2000, dtype=torch.bool, device=device))
    assert all(v.shape[0] == 0 for v in _state(m).values())
