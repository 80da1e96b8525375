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
        for group in self.param_groups:
            self._check_group(group)

    @staticmethod
    def _check_group(group):
        """Everything the kernel cannot do is refused up front (and again at step time for groups added or edited later), BEFORE
        any state is touched."""
        if group.get("amsgrad") or group.get("maximize"):
            raise RuntimeError("FusedAdam: amsgrad / maximize are not implemented")
        if group.get("decoupled_weight_decay"):
            raise RuntimeError("FusedAdam: decoupled_weight_decay (AdamW) is not implemented")
        if isinstance(group["lr"], torch.Tensor):
            raise RuntimeError("FusedAdam: a tensor learning rate would need a device read-back; use a float")
        b1, b2 = group["betas"]
        if not (0.5 < b1 < 1.0) or not (0.0 <= b2 < 1.0):
            raise ValueError(f"FusedAdam: needs 0.5 < beta1 < 1 and 0 <= beta2 < 1 (got {b1}, {b2}): the kernel reproduces torch's foreach "
                             "arithmetic only on that range")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for group in self.param_groups:
            self._check_group(group)
        for group in self.param_groups:      # the whole work list is built and validated before any counter moves
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                if not p.grad.is_contiguous():      # once: the norm and the update read (and the clip writes) the same tensor
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:          # torch.optim.Adam._init_group's layout
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps = {int(self.state[p]["step"]) + 1 for p in ps}
            by_step = {s: [p for p in ps if int(self.state[p]["step"]) + 1 == s] for s in steps}      # normally one entry
            for s, plist in by_step.items():
                work.append((group, s, plist))
        if not work:
            return loss
        clip = None
        if self.max_grad_norm is not None:
            self.last_grad_norm = ops.grad_norm([p.grad for _, _, plist in work for p in plist], self.max_grad_norm)
            clip = self.last_grad_norm[1:]
        for group, s, plist in work:
            grads = [p.grad for p in plist]
            b1, b2 = group["betas"]
            ops.adam_step([p.detach() for p in plist], grads, [self.state[p]["exp_avg"] for p in plist],
                          [self.state[p]["exp_avg_sq"] for p in plist], s, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                          weight_decay=group["weight_decay"], clip_coef=clip, zero_grad=self.zero_grad_in_step)
            for p in plist:
                self.state[p]["step"] += 1    # a CPU tensor: host arithmetic only; advanced only once the launch was accepted
            _bump_versions(plist)
        return loss


class FusedSGD(torch.optim.SGD):
    """torch.optim.SGD (momentum, dampening 0, no Nesterov) with the update of all parameters in one HIP launch.  With
    `max_grad_norm` the clipped gradients are written back to `.grad` (as `clip_grad_norm_` at train.py:165 leaves them),
    unless `zero_grad_in_step` clears them."""

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
        for g, ps in groups:
            if g.get("nesterov") or g.get("dampening") or g.get("maximize"):
                raise RuntimeError("FusedSGD: nesterov / dampening / maximize are not implemented")
            for p in ps:                      # made contiguous ONCE: the norm and the update read (and the clip writes) the same tensor
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
        clip = None
        if self.max_grad_norm is not None:
            self.last_grad_norm = ops.grad_norm([p.grad for _, ps in groups for p in ps], self.max_grad_norm)
            clip = self.last_grad_norm[1:]
        for g, ps in groups:
            mom = g["momentum"]
            fresh = [p for p in ps if mom and "momentum_buffer" not in self.state[p]]
            fresh_ids = {id(p) for p in fresh}
            for first, plist in ((True, fresh), (False, [p for p in ps if id(p) not in fresh_ids])):
                if not plist:
                    continue
                bufs = None
                if mom:
                    for p in plist:
                        if first:
                            self.state[p]["momentum_buffer"] = torch.empty_like(p)
                    bufs = [self.state[p]["momentum_buffer"] for p in plist]
                ops.sgd_step([p.detach() for p in plist], [p.grad for p in plist], bufs, g["lr"], momentum=mom, weight_decay=g["weight_decay"],
                             first_step=first, clip_coef=clip, zero_grad=self.zero_grad_in_step)
                _bump_versions(plist)
        return loss
