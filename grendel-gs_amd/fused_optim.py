"""FusedAdam -- host-side mirror of the optimizer the reference builds at scene/gaussian_model.py:292
(`torch.optim.Adam(l, lr=0.0, eps=1e-15)`, one param group per attribute with its own lr / betas), running
the dense Adam update of ALL parameter tensors as ONE HIP kernel launch (include/gsraster.h: gsr_adam_step_multi) with the
batch-size gradient scaling of train_internal.py:319-324 folded in.

State layout (`state[p]["step" | "exp_avg" | "exp_avg_sq"]`) is the stock optimizer's, so
`GaussianModel.capture()/restore()` checkpoints (scene/gaussian_model.py:70-107) stay interchangeable.
There is no CPU fallback: parameters must live on the gfx950 device.

`fuse_backward=True` (opt-in) additionally fuses the projection backward K11 into the step: `loss.backward()`
(train_internal.py:195) then stops in front of K11 -- the six raw parameters get NO `.grad` -- and `step()` runs K11 and
the Adam update of those six tensors as one kernel, so their gradients (236 B per Gaussian, written by K11 and read back
by the optimizer otherwise) never touch HBM.  The arithmetic is the unfused pair's, bit for bit.  Whatever happens
between backward and step in the reference's loop keeps its meaning: densification replaces parameters (their pending
gradient is dropped, exactly like the `.grad is None` skip of the stock optimizer, or -- for a partial replacement such
as reset_opacity -- materialized for the tensors that are still current), `zero_grad()` without a step drops it, a second
backward before the step materializes both.  What changes: code that READS `.grad` of the raw parameters between
backward and step sees None -- the reference's own `param.grad /= bsz` loop (train_internal.py:319-324) would be such a
reader and silently do nothing, so the scale has to be passed as `grad_scale` (constructor or step()), as bench.py
does.  `means2D.grad` (densification statistics) is unaffected: it comes from K10, not K11."""
import ctypes

import torch

import diff_gaussian_rasterization as _dgr
from diff_gaussian_rasterization import _lib, _on, _stream, kernel_timer


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fuse_backward=False, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = float(grad_scale)
        self._pending = None
        self._owners_cache = None
        self._launch_cache = {}
        self._warned_replicated = False
        self.fused_steps = 0         # steps in which K11 ran inside the optimizer kernel
        self.materialized_steps = 0  # deferred backwards that had to fall back to the plain K11
        self.fuse_backward = False
        if fuse_backward:
            self.set_fuse_backward(True)

    def __setstate__(self, state):
        super().__setstate__(state)
        # an unpickled optimizer never is the sink of a pending backward: re-arm with set_fuse_backward(True)
        self.__dict__.setdefault("grad_scale", 1.0)
        self._pending, self._owners_cache, self._launch_cache, self.fuse_backward = None, None, {}, False
        self._warned_replicated = False
        self.__dict__.setdefault("fused_steps", 0)
        self.__dict__.setdefault("materialized_steps", 0)

    # ------------------------------------------------------------------ deferred K11 (see the module docstring)
    def set_fuse_backward(self, on):
        """register / unregister this optimizer as the sink of the operator's deferred projection backward (one
        optimizer per process can be the sink)"""
        self.fuse_backward = bool(on)
        if on:
            _dgr.set_deferred_backward_sink(self)
        elif _dgr.deferred_backward_sink() is self:
            self._flush_pending()
            _dgr.set_deferred_backward_sink(None)

    def _owner(self, t):
        """(group, parameter) of this optimizer whose storage is tensor t, or None"""
        ptr, shape = t.data_ptr(), t.shape
        for group in self.param_groups:
            for p in group["params"]:
                if p.data_ptr() == ptr and p.shape == shape and p.dtype == t.dtype:
                    return group, p
        return None

    def _owners_of(self, params):
        """[(group, parameter)] x 6 for the six tensors of a projection backward, or None when one of them is not
        optimized here (or two share a parameter); cached per set of addresses -- the lookup runs every iteration"""
        key = tuple(t.data_ptr() for t in params)
        cached = self._owners_cache
        if cached is not None and cached[0] == key and \
                all(any(q is p for q in g["params"]) and p.data_ptr() == k for (g, p), k in zip(cached[1], key)):
            return cached[1]
        owners = [self._owner(t) for t in params]
        if any(o is None for o in owners) or len({id(o[1]) for o in owners}) != len(owners):
            self._owners_cache = None
            return None
        self._owners_cache = (key, owners)
        return owners

    def accepts(self, params):
        """called by the operator's backward: True when all six raw parameters are optimized here and carry no other
        gradient -- then offer() follows and the node returns no gradients"""
        if not self.fuse_backward:
            return False
        owners = self._owners_of(params)
        if owners is None or not all(p.grad is None and p.requires_grad for _, p in owners):
            return False
        if not self._warned_replicated:
            # replicated storage (world > 1 without --gaussians_distribution) all-reduces `.grad` between backward and
            # step (scene/gaussian_model.py: sync_gradients_for_replicated_3dgs_storage): a deferred backward leaves
            # `.grad` None and that sync would silently do nothing
            try:
                import utils.general_utils as utils

                args = utils.get_args()
                if utils.DEFAULT_GROUP is not None and utils.DEFAULT_GROUP.size() > 1 and args is not None and \
                        not getattr(args, "gaussians_distribution", True):
                    import warnings

                    warnings.warn("FusedAdam(fuse_backward=True) with replicated Gaussian storage: the gradient "
                                  "all-reduce between backward and step sees no `.grad`; use fuse_backward=False")
            except Exception:  # noqa: BLE001  (no mirror `utils` on the path: nothing to check)
                pass
            self._warned_replicated = True
        return True

    def offer(self, pending):
        if self._pending is not None:  # a second backward before the step: both become ordinary gradients
            self._flush_pending()
            self._pending = pending
            self._flush_pending()
            return
        self._pending = pending

    def _flush_pending(self):
        """materialize the pending projection backward into `.grad` of the parameters that are still the tensors it
        was computed for (a replaced parameter has no gradient, as in the stock flow)"""
        pend, self._pending = self._pending, None
        if pend is None:
            return
        grads = None
        for idx, t in enumerate(pend.params):
            o = self._owner(t)
            if o is None:
                continue
            if t._version != pend.versions[idx]:
                raise RuntimeError("FusedAdam(fuse_backward=True): a parameter was modified in place between "
                                   "backward() and step(); its deferred gradient would be computed from the new value")
            if grads is None:
                grads = pend.materialize()
                self.materialized_steps += 1
            p, g = o[1], grads[idx].view(o[1].shape)
            p.grad = g if p.grad is None else p.grad.add_(g)

    def zero_grad(self, set_to_none=True):
        self._pending = None  # a pending projection backward is a gradient too
        return super().zero_grad(set_to_none=set_to_none)

    def _fused_backward_step(self, grad_scale):
        """-> set of parameters updated by the fused K11 + Adam launch (empty when the pending backward had to be
        materialized instead)"""
        pend = self._pending
        owners = self._owners_of(pend.params)  # None: a parameter was replaced since the backward
        fusable = owners is not None and all(p.grad is None and p.is_contiguous() for _, p in owners)
        if fusable:
            for idx, t in enumerate(pend.params):
                if t._version != pend.versions[idx]:
                    raise RuntimeError("FusedAdam(fuse_backward=True): a parameter was modified in place between "
                                       "backward() and step(); its deferred gradient would be computed from the new "
                                       "value")
        if not fusable:
            self._flush_pending()
            return set()
        self._pending = None
        sts = []
        for group, p in owners:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            sts.append(st)
        cap = _dgr.capturing()
        if cap is not None:
            # a captured iteration at world size > 1: the capacity flag is raised per rank (a band of THIS rank received too
            # few rows, THIS rank's pair count outgrew its sort capacity), but skipping the update must be ONE decision --
            # a rank that skips repeats the iteration eagerly and its collectives would pair with its peers' next replays.
            # One 4-byte all-reduce (MAX) in front of the update makes every rank see every rank's flag
            import torch.distributed as dist

            import utils.general_utils as utils

            group = utils.DEFAULT_GROUP
            if group is not None and group.size() > 1 and dist.is_initialized():
                dist.all_reduce(cap.flag, op=dist.ReduceOp.MAX,
                                group=group if isinstance(group, dist.ProcessGroup) else None)
        # (the launch first, the host-side step counters after it: at world size > 1 the host is the bottleneck and
        # everything in front of this launch is time the GPU idles)
        pend.fused_step([st["exp_avg"] for st in sts], [st["exp_avg_sq"] for st in sts],
                        [g["lr"] for g, _ in owners], [g["betas"][0] for g, _ in owners],
                        [g["betas"][1] for g, _ in owners], [g["eps"] for g, _ in owners],
                        [int(st["step"]) + 1 for st in sts], grad_scale, cache=self._launch_cache)
        if _dgr.capturing() is not None:
            # the launch was recorded into a hipGraph, not executed: the step counters move when a replay is launched
            # (graph_advance); the graph's owner finds the six (group, parameter) pairs here
            self._graph_owners = owners
            return {id(p) for _, p in owners}
        torch._foreach_add_([st["step"] for st in sts], 1)
        self.fused_steps += 1
        return {id(p) for _, p in owners}

    # ------------------------------------------------------------------ a captured iteration (graphed_step.py)
    def graph_hyper(self):
        """the 12 per-step constants of the captured fused launch for the NEXT step, from the param groups' current
        learning rates and the step counters: lr / (1 - beta1^t) x 6, then 1 / sqrt(1 - beta2^t) x 6, in the order of
        the six tensors of the projection backward (include/gsraster.h: gsr_preprocess_backward_adam_raw_batched_dyn)"""
        out = [0.0] * 12
        for t, (group, p) in enumerate(self._graph_owners):
            step = int(self.state[p]["step"]) + 1
            b1, b2 = group["betas"]
            out[t] = group["lr"] / (1.0 - b1 ** step)
            out[6 + t] = 1.0 / (1.0 - b2 ** step) ** 0.5
        return out

    def graph_advance(self, n=1):
        """a replay of the captured iteration has been launched (n > 0) / n replays turned out to be no-ops (n < 0)"""
        torch._foreach_add_([self.state[p]["step"] for _, p in self._graph_owners], float(n))
        self.fused_steps += n

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """`grad_scale` multiplies every gradient before the update (1 / bsz in the reference's loop; default: the
        constructor's).  All parameter tensors that have a gradient are updated by ONE kernel launch
        (gsr_adam_step_multi); with fuse_backward the six raw parameters of a pending projection backward are updated
        by the fused K11 + Adam launch instead."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if grad_scale is None:
            grad_scale = self.grad_scale
        done = self._fused_backward_step(grad_scale) if self._pending is not None else ()
        batch, steps = [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            for p in group["params"]:
                g = p.grad
                if g is None or id(p) in done:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must live on the gfx950 device (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and
                        g.is_contiguous()):
                    raise RuntimeError("FusedAdam: dense fp32 parameters and gradients expected")
                steps.append(st["step"])
                batch.append((p, g, st, lr, b1, b2, eps))
        if batch and _dgr.capturing() is not None:
            raise RuntimeError("FusedAdam: a captured iteration supports the fused K11 + Adam launch only (parameters "
                               "outside the projection backward carry gradients)")
        if steps:
            torch._foreach_add_(steps, 1)  # host tensors, as in the stock optimizer's state
        for i in range(0, len(batch), 16):
            part = batch[i:i + 16]
            K = len(part)
            dev = part[0][0].device
            # the pointer tables of parameters and moments only change when the model is rebuilt (densification,
            # restore): cached per launch group, keyed by the parameters' addresses
            # AND by everything else the plan holds (sizes, both moments): densification replaces every tensor several
            # times between two steps and the caching allocator may hand an old address back with a different size
            key = tuple((b[0].data_ptr(), b[0].numel(), b[2]["exp_avg"].data_ptr(), b[2]["exp_avg_sq"].data_ptr())
                        for b in part)
            plan = self._plans.get(i) if hasattr(self, "_plans") else None
            if plan is None or plan[0] != key:
                for b in part:
                    if b[0].device != dev:
                        raise RuntimeError("FusedAdam: all parameters of one step must live on one device")
                VP, I64 = ctypes.c_void_p * K, ctypes.c_int64 * K
                numels = [b[0].numel() for b in part]
                plan = (key, I64(*numels), VP(*[b[0].data_ptr() for b in part]),
                        VP(*[b[2]["exp_avg"].data_ptr() for b in part]),
                        VP(*[b[2]["exp_avg_sq"].data_ptr() for b in part]), sum(numels))
                if not hasattr(self, "_plans"):
                    self._plans = {}
                self._plans[i] = plan
            VP, D, I64 = ctypes.c_void_p * K, ctypes.c_double * K, ctypes.c_int64 * K
            with _on(dev), kernel_timer.range("adam", numel=plan[5]):
                _lib.check(_lib.lib.gsr_adam_step_multi(
                    K, plan[1], plan[2], VP(*[b[1].data_ptr() for b in part]), plan[3], plan[4],
                    D(*[b[3] for b in part]), D(*[b[4] for b in part]), D(*[b[5] for b in part]),
                    D(*[b[6] for b in part]), I64(*[int(b[2]["step"]) for b in part]), float(grad_scale),
                    _stream()), "gsr_adam_step_multi")
        return loss
