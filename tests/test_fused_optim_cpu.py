"""CPU: the bookkeeping of FusedAdam(fuse_backward=True) -- which backward passes are deferred, what step(),
zero_grad(), a second backward and a replaced parameter do with a pending projection backward -- with a stand-in for
the operator's PendingProjectionBackward (the kernels themselves: tests/test_gpu_loss_and_step.py)."""
import pytest
import torch

import synthetic_scene as S

NAMES = ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")


@pytest.fixture(autouse=True)
def _no_sink_left_behind():
    import diff_gaussian_rasterization as dgr

    yield
    dgr.set_deferred_backward_sink(None)


class FakePending:
    def __init__(self, params):
        self.params = params
        self.versions = tuple(t._version for t in params)
        self.fused_calls = []
        self.materialized = 0

    def materialize(self):
        self.materialized += 1
        return tuple(torch.full_like(t, float(i + 1)) for i, t in enumerate(self.params))

    def fused_step(self, exp_avgs, exp_avg_sqs, lrs, b1, b2, eps, steps, grad_scale, cache=None):
        self.fused_calls.append((list(lrs), list(steps), grad_scale, [tuple(t.shape) for t in exp_avgs]))


def make():
    from fused_optim import FusedAdam

    m = S.SyntheticGaussianModel(64, 64, 48, seed=1)
    opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15, fuse_backward=True, grad_scale=0.25)
    return m, opt, tuple(getattr(m, n) for n in NAMES)


def test_sink_registration_and_acceptance():
    import diff_gaussian_rasterization as dgr
    from fused_optim import FusedAdam

    m, opt, params = make()
    try:
        assert dgr.deferred_backward_sink() is opt and opt.accepts(params)
        other = S.SyntheticGaussianModel(64, 64, 48, seed=2)
        assert not opt.accepts(tuple(getattr(other, n) for n in NAMES))  # somebody else's parameters
        m._xyz.grad = torch.zeros_like(m._xyz)
        assert not opt.accepts(params)  # another gradient path already wrote .grad: do not defer
        m._xyz.grad = None
        plain = FusedAdam(other.param_groups(), lr=0.0, eps=1e-15)
        assert dgr.deferred_backward_sink() is opt and not plain.accepts(tuple(getattr(other, n) for n in NAMES))
        opt.set_fuse_backward(False)
        assert dgr.deferred_backward_sink() is None and not opt.accepts(params)
        # the sink is held weakly: an optimizer nobody references any more cannot swallow a backward
        opt.set_fuse_backward(True)
        assert dgr.deferred_backward_sink() is opt
        import gc

        del opt
        gc.collect()
        assert dgr.deferred_backward_sink() is None
        opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15)
    finally:
        opt.set_fuse_backward(False)


def test_step_hands_the_six_groups_to_the_fused_launch():
    m, opt, params = make()
    try:
        pend = FakePending(params)
        opt.offer(pend)
        opt.step()
        assert opt._pending is None and opt.fused_steps == 1 and pend.materialized == 0
        (lrs, steps, gs, shapes), = pend.fused_calls
        # tensor order of the kernel (xyz, scaling, rotation, f_dc, f_rest, opacity) with each group's own lr
        assert lrs == [0.00016, 0.005, 0.001, 0.0025, 0.0025 / 20.0, 0.05]
        assert steps == [1] * 6 and gs == 0.25
        assert shapes == [tuple(p.shape) for p in params]
        assert all(float(opt.state[p]["step"]) == 1.0 for p in params)
        opt.offer(FakePending(params))
        opt.step(grad_scale=0.5)
        assert all(float(opt.state[p]["step"]) == 2.0 for p in params)
    finally:
        opt.set_fuse_backward(False)


def test_zero_grad_drops_and_second_backward_materializes():
    m, opt, params = make()
    try:
        pend = FakePending(params)
        opt.offer(pend)
        opt.zero_grad(set_to_none=True)
        assert opt._pending is None and pend.materialized == 0
        a, b = FakePending(params), FakePending(params)
        opt.offer(a)
        opt.offer(b)  # gradient accumulation: both become ordinary, summed `.grad`s
        assert opt._pending is None and a.materialized == 1 and b.materialized == 1
        for i, p in enumerate(params):
            assert torch.equal(p.grad, torch.full_like(p, 2.0 * (i + 1)))
        assert not opt.accepts(params)  # .grad is set now: the next backward accumulates the stock way
    finally:
        opt.set_fuse_backward(False)


def test_replaced_parameter_is_skipped_and_inplace_edit_is_refused():
    m, opt, params = make()
    try:
        pend = FakePending(params)
        opt.offer(pend)
        new = torch.nn.Parameter(m._opacity.detach().clone())  # reset_opacity / densification: a new tensor in the group
        for g in opt.param_groups:
            if g["name"] == "opacity":
                g["params"][0] = new
        with pytest.raises(RuntimeError, match="gfx950"):  # five ordinary gradients on host tensors: no CPU optimizer
            opt.step()
        assert pend.materialized == 1 and not pend.fused_calls and new.grad is None
        assert all(getattr(m, n).grad is not None for n in NAMES if n != "_opacity")

        m2, opt2, params2 = make()
        opt2.offer(FakePending(params2))
        with torch.no_grad():
            m2._xyz.add_(1.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            opt2.step()
    finally:
        opt.set_fuse_backward(False)


def test_unpickled_optimizer_comes_back_disarmed():
    """__setstate__ (what unpickling / copy.deepcopy of an optimizer runs): no pending backward, not the sink"""
    m, opt, params = make()
    opt.offer(FakePending(params))
    fresh = opt.__class__.__new__(opt.__class__)
    fresh.__setstate__({"defaults": opt.defaults, "state": {}, "param_groups": opt.param_groups})
    assert fresh._pending is None and fresh.fuse_backward is False and fresh.grad_scale == 1.0
    assert not fresh.accepts(params)


@pytest.fixture
def fake_library(monkeypatch):
    """the operator's Python plumbing on HOST tensors: every C-ABI entry point is replaced by a ctypes callback with the
    real prototype (include/gsraster.h via _lib.SIGNATURES), so argument counts and types are checked exactly as on
    the GPU box while nothing is computed; -> list of (entry point, number of arguments) in call order"""
    import contextlib
    import ctypes

    import diff_gaussian_rasterization as dgr
    import fused_optim
    from diff_gaussian_rasterization import _lib

    calls = []

    class Fake:
        def __getattr__(self, name):
            res, args = _lib.SIGNATURES[name]

            def cb(*a):
                calls.append((name, len(a)))
                return 0

            f = ctypes.CFUNCTYPE(res, *args)(cb)
            setattr(self, name, f)
            return f

    fake = Fake()
    monkeypatch.setattr(dgr, "lib", fake)
    monkeypatch.setattr(fused_optim._lib, "lib", fake)
    monkeypatch.setattr(dgr, "_f32c", lambda t, n: t.float().contiguous())
    monkeypatch.setattr(dgr, "_on", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(dgr, "_stream", lambda: None)
    return calls


@pytest.mark.parametrize("B", [1, 3])
def test_operator_hands_the_projection_backward_to_the_optimizer(fake_library, B):
    """_PreprocessGaussiansRawBatched.backward with a sink: no K11 launch, no `.grad`, ONE fused launch at step();
    without a sink (or after set_fuse_backward(False)): the plain K11 of that batch size; two backwards before one
    step: two plain K11 launches, summed gradients"""
    import diff_gaussian_rasterization as dgr
    from fused_optim import FusedAdam

    calls = fake_library
    m = S.SyntheticGaussianModel(500, 64, 48, seed=1)
    opt = FusedAdam(m.param_groups(), lr=0.0, eps=1e-15, fuse_backward=True, grad_scale=1.0 / B)
    raw = [getattr(m, n) for n in NAMES]
    cams = torch.zeros(B, 40)

    def backward_once():
        m2, rgb, co, radii, depths = dgr.preprocess_gaussians_raw_batched(*raw, cams, 3, 1.0, 64, 48, tanfov0=(0.5, 0.4))
        sum((m2[k] ** 2).sum() + rgb[k].sum() + co[k].sum() for k in range(B)).backward()

    backward_once()
    assert all(p.grad is None for p in raw) and opt._pending is not None
    assert [c[0] for c in calls] == ["gsr_preprocess_forward_raw_batched"]
    opt.step()
    opt.zero_grad()
    assert [c[0] for c in calls][-1] == "gsr_preprocess_backward_adam_raw_batched" and opt.fused_steps == 1
    assert all(float(opt.state[p]["step"]) == 1.0 for p in raw)
    calls.clear()
    opt.set_fuse_backward(False)
    backward_once()
    plain = "gsr_preprocess_backward_raw" if B == 1 else "gsr_preprocess_backward_raw_batched"
    assert [c[0] for c in calls] == ["gsr_preprocess_forward_raw_batched", plain]
    assert all(p.grad is not None for p in raw)
    opt.zero_grad()
    calls.clear()
    opt.set_fuse_backward(True)
    backward_once()
    backward_once()
    assert [c[0] for c in calls].count(plain) == 2 and opt._pending is None and opt.materialized_steps == 2
    assert all(p.grad is not None for p in raw)
