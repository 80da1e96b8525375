"""Optimisers of the reference's train.py (Adam lr 1e-3: train.py:242; SGD momentum 0.9: train.py:244) as ONE fused
multi-tensor launch per step, with `clip_grad_norm_` (train.py:165) folded in -- SURVEY 8f rank 3.

Both classes subclass their torch.optim counterparts and keep torch's state layout (`state[p]['step'|'exp_avg'|'exp_avg_sq']`,
`state[p]['momentum_buffer']`), so `optimizer.state_dict()` / `load_state_dict()` -- the `optimizer` entry of the reference's
checkpoint dict (train.py:283-301) -- interchange with torch.optim.Adam / SGD checkpoints, and LR schedulers
(StepLR, train.py:251-252) drive `param_groups[...]['lr']` as usual.  The arithmetic follows torch's foreach implementation
operation by operation; the step counter lives on the host (a CPU tensor, as torch keeps it), so a step performs no device
read-back, no H2D copy and no host-side tensor math.
"""
import torch

from . import ops


def _bump_versions(params):
    """The kernels wrote the parameters through raw pointers: bump their autograd version counters so that everything keyed on
    `_version` (the packed-filter caches of model.Conv2DBlock, autograd's saved-tensor checks) sees the update."""
    if hasattr(torch._C, "_increment_version"):
        torch._C._increment_version(list(params))
    else:
        for p in params:
            p.add_(0)


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam with the update of all parameters in one HIP launch.  max_grad_norm: clip the GLOBAL gradient norm like
    `torch.nn.utils.clip_grad_norm_(params, max_grad_norm)` right before the update (two extra small launches; the clipped
    gradients are written back to `.grad`).  zero_grad_in_step: also zero the gradients inside the same launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, max_grad_norm=None, zero_grad_in_step=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, capturable=False,
                         fused=False)
        self.max_grad_norm, self.zero_grad_in_step = max_grad_norm, bool(zero_grad_in_step)
        self.last_grad_norm = None            # (2,) device tensor [norm, clip coefficient] of the last clipped step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("FusedAdam: amsgrad / maximize are not implemented")
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                raise RuntimeError("FusedAdam: a tensor learning rate would need a device read-back; use a float")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if len(st) == 0:          # torch.optim.Adam._init_group's layout
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1           # a CPU tensor: host arithmetic only
            steps = {int(self.state[p]["step"]) for p in ps}
            by_step = {s: [p for p in ps if int(self.state[p]["step"]) == s] for s in steps}      # normally one entry
            for s, plist in by_step.items():
                work.append((group, s, plist))
        if not work:
            return loss
        clip = None
        if self.max_grad_norm is not None:
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for _, _, plist in work for p in plist]
            self.last_grad_norm = ops.grad_norm(grads, self.max_grad_norm)
            clip = self.last_grad_norm[1:]
        for group, s, plist in work:
            grads = []
            for p in plist:
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                grads.append(p.grad)
            b1, b2 = group["betas"]
            ops.adam_step([p.detach() for p in plist], grads, [self.state[p]["exp_avg"] for p in plist],
                          [self.state[p]["exp_avg_sq"] for p in plist], s, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                          weight_decay=group["weight_decay"], clip_coef=clip, zero_grad=self.zero_grad_in_step)
            _bump_versions(plist)
        return loss


class FusedSGD(torch.optim.SGD):
    """torch.optim.SGD (momentum, dampening 0, no Nesterov) with the update of all parameters in one HIP launch."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0, max_grad_norm=None, zero_grad_in_step=False):
        super().__init__(params, lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False, foreach=False)
        self.max_grad_norm, self.zero_grad_in_step = max_grad_norm, bool(zero_grad_in_step)
        self.last_grad_norm = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        groups = [(g, ps) for g, ps in groups if ps]
        if not groups:
            return loss
        clip = None
        if self.max_grad_norm is not None:
            self.last_grad_norm = ops.grad_norm([p.grad.contiguous() for _, ps in groups for p in ps], self.max_grad_norm)
            clip = self.last_grad_norm[1:]
        for g, ps in groups:
            if g.get("nesterov") or g.get("dampening") or g.get("maximize"):
                raise RuntimeError("FusedSGD: nesterov / dampening / maximize are not implemented")
            mom = g["momentum"]
            fresh = [p for p in ps if mom and "momentum_buffer" not in self.state[p]]
            for first, plist in ((True, fresh), (False, [p for p in ps if p not in set(fresh)])):
                if not plist:
                    continue
                bufs = None
                if mom:
                    for p in plist:
                        if first:
                            self.state[p]["momentum_buffer"] = torch.empty_like(p)
                    bufs = [self.state[p]["momentum_buffer"] for p in plist]
                for p in plist:
                    if not p.grad.is_contiguous():
                        p.grad = p.grad.contiguous()
                ops.sgd_step([p.detach() for p in plist], [p.grad for p in plist], bufs, g["lr"], momentum=mom, weight_decay=g["weight_decay"],
                             first_step=first, clip_coef=clip, zero_grad=self.zero_grad_in_step)
                _bump_versions(plist)
        return loss
