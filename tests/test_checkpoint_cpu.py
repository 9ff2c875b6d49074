"""N4, checkpoint IO: the reference's own re-sharding loaders (utils/general_utils.py:516-709:
load_checkpoint -> merge_multiple_checkpoints / get_part_of_checkpoints / drop_duplicate_gaussians) run UNCHANGED on
checkpoints written from this package's model / optimizer objects -- GaussianModel.capture()-shaped tuples
(scene/gaussian_model.py:70-84) with FusedAdam's state_dict in the optimizer slot -- for W = 2 -> 1, 1 -> 2 and 2 -> 2,
and FusedAdam / stock torch.optim.Adam state dicts are interchangeable.  Needs the reference tree (build container
only); the loaders hard-code map_location="cuda:<rank>", which this GPU-less container cannot honour, so torch.load is
wrapped to map to the CPU -- the re-sharding arithmetic under test is device independent."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PKG = os.path.join(ROOT, "grendel-gs_amd")


def _fabricated_state(opt):
    """what FusedAdam.step() leaves in optimizer.state (fused_optim.py), without a GPU"""
    for gi, group in enumerate(opt.param_groups):
        for p in group["params"]:
            opt.state[p] = {"step": torch.tensor(7.0), "exp_avg": torch.full_like(p, 0.25 + gi),
                            "exp_avg_sq": torch.full_like(p, 0.5 + gi)}


def test_fused_adam_state_dict_is_the_stock_optimizers():
    sys.path.insert(0, PKG)
    import synthetic_scene as S
    from fused_optim import FusedAdam

    m = S.SyntheticGaussianModel(64, 64, 48, seed=1)
    fa = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    _fabricated_state(fa)
    sd = fa.state_dict()
    m2 = S.SyntheticGaussianModel(64, 64, 48, seed=1)
    stock = torch.optim.Adam(m2.param_groups(), lr=0.0, eps=1e-15)
    stock.load_state_dict(sd)  # FusedAdam -> stock
    assert [g["name"] for g in stock.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert float(stock.state[m2._opacity]["exp_avg"].mean()) == 3.25 and float(stock.state[m2._xyz]["step"]) == 7.0
    fb = FusedAdam(S.SyntheticGaussianModel(64, 64, 48, seed=1).param_groups(), lr=0.0, eps=1e-15)
    fb.load_state_dict(stock.state_dict())  # stock -> FusedAdam
    p = fb.param_groups[3]["params"][0]
    assert float(fb.state[p]["exp_avg_sq"].mean()) == 3.5 and fb.param_groups[3]["lr"] == 0.05


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "utils", "general_utils.py")), reason="reference tree not present")
def test_reference_resharding_loaders_on_our_checkpoints(tmp_path):
    code = r'''
import os, sys, types, torch
sys.path.insert(0, %(pkg)r)
import synthetic_scene as S
from fused_optim import FusedAdam
sys.path.remove(%(pkg)r)
for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
    del sys.modules[k]
sys.path.insert(0, %(ref)r)
import utils.general_utils as u          # the REFERENCE's module
assert u.__file__.startswith(%(ref)r)
_load = torch.load
torch.load = lambda f, map_location=None, **kw: _load(f, map_location="cpu", weights_only=False, **kw)

N, W, H = 1001, 64, 48
class G:
    def __init__(s, n, r): s.n, s.r = n, r
    def size(s): return s.n
    def rank(s): return s.r

def capture(model, opt, it):   # scene/gaussian_model.py:70-84
    n = model._xyz.shape[0]
    return ((model.active_sh_degree, model._xyz, model._features_dc, model._features_rest, model._scaling,
             model._rotation, model._opacity, torch.arange(n, dtype=torch.float32), torch.ones(n, 1), torch.ones(n, 1),
             opt.state_dict(), 1.5), it)

def shard(rank, world):
    m = S.SyntheticGaussianModel(N, W, H, seed=3, rank=rank, world_size=world)
    o = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    for gi, group in enumerate(o.param_groups):
        for p in group["params"]:
            o.state[p] = {"step": torch.tensor(7.0), "exp_avg": torch.full_like(p, 0.25 + gi), "exp_avg_sq": torch.full_like(p, 0.5)}
    return m, o

full, _ = shard(0, 1)
d2 = %(tmp)r + "/ws2/"; d1 = %(tmp)r + "/ws1/"
os.makedirs(d2); os.makedirs(d1)
for r in range(2):
    m, o = shard(r, 2)
    torch.save(capture(m, o, 3000), d2 + "chkpnt_ws=2_rk=%%d.pth" %% r)   # train_internal.py's file naming
m, o = shard(0, 1)
torch.save(capture(m, o, 3000), d1 + "chkpnt_ws=1_rk=0.pth")
args = types.SimpleNamespace(start_checkpoint=d2, drop_duplicate_gaussians_coeff=1.0)

# W = 2 -> 1: both files merged on the single rank
u.DEFAULT_GROUP, u.LOCAL_RANK = G(1, 0), 0
params, it = u.load_checkpoint(args)
assert it == 3000 and params[10] is None and params[0] == 3 and params[11] == 1.5
for idx, name in ((1, "_xyz"), (2, "_features_dc"), (3, "_features_rest"), (4, "_scaling"), (5, "_rotation"), (6, "_opacity")):
    assert torch.equal(params[idx].detach(), getattr(full, name).detach()), name
    assert isinstance(params[idx], torch.nn.Parameter) and params[idx].requires_grad
assert params[7].shape == (N,) and params[8].shape == (N, 1)
# ... and the merged tuple feeds this package's optimizer (what GaussianModel.restore + training_setup do)
mm = S.SyntheticGaussianModel(8, W, H, seed=0)
mm._xyz, mm._features_dc, mm._features_rest, mm._scaling, mm._rotation, mm._opacity = params[1:7]
FusedAdam(mm.param_groups(), lr=0.0, eps=1e-15)

# W = 2 -> 2: every rank reads its own file, optimizer state kept and loadable by FusedAdam and by stock Adam
args.start_checkpoint = d2
for r in range(2):
    u.DEFAULT_GROUP = G(2, r)
    params, it = u.load_checkpoint(args)
    ms, _ = shard(r, 2)
    assert torch.equal(params[1].detach(), ms._xyz.detach())
    for cls in (FusedAdam, torch.optim.Adam):
        mt, _ = shard(r, 2)
        opt = cls(mt.param_groups(), lr=0.0, eps=1e-15)
        opt.load_state_dict(params[10])
        assert float(opt.state[mt._opacity]["exp_avg"].mean()) == 3.25

# W = 1 -> 2: each rank takes its part of the single file (the reference's split: N // parts + 1 rows per part)
args.start_checkpoint = d1
got = []
for r in range(2):
    u.DEFAULT_GROUP = G(2, r)
    params, it = u.load_checkpoint(args)
    got.append(params[1].detach())
    assert params[10] is None
per = N // 2 + 1
assert got[0].shape[0] == per and got[1].shape[0] == N - per
assert torch.equal(torch.cat(got), full._xyz.detach())
# duplicated Gaussians (drop_duplicate_gaussians_coeff > 1 tiles the rows)
args.drop_duplicate_gaussians_coeff = 1.5
u.DEFAULT_GROUP = G(1, 0)
params, _ = u.load_checkpoint(args)
assert params[1].shape[0] == int(N * 1.5) and torch.equal(params[1][N:].detach(), full._xyz[: int(N * 1.5) - N].detach())
print("ok")
''' % {"pkg": PKG, "ref": REF, "tmp": str(tmp_path)}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=""))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
