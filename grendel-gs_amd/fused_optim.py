"""FusedAdam -- host-side mirror of the optimizer the reference builds at scene/gaussian_model.py:292
(`torch.optim.Adam(l, lr=0.0, eps=1e-15)`, one param group per attribute with its own lr / betas), running
the dense Adam update of ALL parameter tensors as ONE HIP kernel launch (include/gsraster.h: gsr_adam_step_multi) with the
batch-size gradient scaling of train_internal.py:319-324 folded in.

State layout (`state[p]["step" | "exp_avg" | "exp_avg_sq"]`) is the stock optimizer's, so
`GaussianModel.capture()/restore()` checkpoints (scene/gaussian_model.py:70-107) stay interchangeable.
There is no CPU fallback: parameters must live on the gfx950 device."""
import ctypes

import torch

from diff_gaussian_rasterization import _lib, _on, _stream, kernel_timer


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """`grad_scale` multiplies every gradient before the update (1 / bsz in the reference's loop).  All parameter
        tensors that have a gradient are updated by ONE kernel launch (gsr_adam_step_multi)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batch, steps = [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            for p in group["params"]:
                g = p.grad
                if g is None:
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
        if steps:
            torch._foreach_add_(steps, 1)  # host tensors, as in the stock optimizer's state
        for i in range(0, len(batch), 16):
            part = batch[i:i + 16]
            K = len(part)
            dev = part[0][0].device
            # the pointer tables of parameters and moments only change when the model is rebuilt (densification,
            # restore): cached per launch group, keyed by the parameters' addresses
            key = tuple(b[0].data_ptr() for b in part) + tuple(b[2]["exp_avg"].data_ptr() for b in part)
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
