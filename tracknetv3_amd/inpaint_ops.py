"""InpaintNet.forward (model.py:113-129) as nine conv1d launches of libtnv3_hip.so; concats are two-pointer reads."""
import torch

from . import ops


# The fused single-kernel forward (csrc/kernels/inpaint_fused.h) is the default for sequences of 16 positions: measured on MI355X
# (scripts/microbench.py, profiles/r02_microbench.json) it is faster than the nine per-layer launches at every batch size --
# 0.054 vs 0.119 ms at the README batch of 32 (both dominated by host launch latency), 0.059 vs 0.140 ms at 256, 0.33 vs 0.52 ms
# at 2048 and 6.32 vs 6.28 M sequences/s at 65 536.  TNV3_INPAINT_FUSED=0 forces the per-layer kernels (other sequence lengths
# and training always use them).
import os

FUSED = os.environ.get("TNV3_INPAINT_FUSED", "auto")


def packed_params(net):
    """The fused kernel's parameter buffer, rebuilt when any of the 18 tensors changes (version counters)."""
    params = net.conv_params()
    key = tuple((t.data_ptr(), t._version) for wb in params for t in wb)
    hit = getattr(net, "_fused_pack", None)
    if hit is None or hit[0] != key:
        hit = (key, ops.inpaintnet_pack([w.detach() for w, _ in params], [b.detach() for _, b in params]))
        net._fused_pack = hit
    return hit[1]


def packed_params_t(net):
    """The fused backward's transposed filter buffer, rebuilt when a weight changes (version counters)."""
    params = net.conv_params()
    key = tuple((w.data_ptr(), w._version) for w, _ in params)
    hit = getattr(net, "_fused_pack_t", None)
    if hit is None or hit[0] != key:
        hit = (key, ops.inpaintnet_pack_t([w.detach() for w, _ in params]))
        net._fused_pack_t = hit
    return hit[1]


# The training step's forward + backward as three launches (csrc/kernels/inpaint_fused_train.h) instead of ~35; TNV3_INPAINT_FUSED_TRAIN=0
# (or another sequence length than 16) keeps the per-layer kernels.  Measured at the README batch of 32 (bench.py `inpaintnet.train_n32`):
# 1.06 -> 0.47 ms per step incl. the masked MSE, clip_grad_norm_ and Adam.  The all-layer weight-gradient launch walks the batch
# sequentially per (16 x 16)-channel item -- right for training batches, wrong for huge ones (one fp32 chain over 600 000 sequences
# loses 1e-3 and takes seconds), so batches beyond FUSED_TRAIN_MAX_BATCH use the per-layer kernels with their split reductions.
FUSED_TRAIN = os.environ.get("TNV3_INPAINT_FUSED_TRAIN", "1")
FUSED_TRAIN_MAX_BATCH = 4096


def use_fused_train(n, seq_len):
    return seq_len == 16 and FUSED_TRAIN != "0" and FUSED != "0" and 0 < n <= FUSED_TRAIN_MAX_BATCH


def use_fused(n, seq_len):
    return seq_len == 16 and FUSED != "0"


def inpaintnet_forward(net, x, m):
    if net.training and torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
        from . import autograd_ops
        return autograd_ops.inpaintnet_forward_train(net, x, m)
    with torch.no_grad():
        x = x.contiguous().float()
        m = m.contiguous().to(torch.float32)
        if use_fused(int(x.shape[0]), int(x.shape[1])):
            return ops.inpaintnet_fused(x, m, packed_params(net))
        p = [(w.detach(), b.detach()) for w, b in net.conv_params()]
        x1 = ops.conv1d_k3(x, *p[0], src1=m, src_nlc=True)                 # cat([x, m], 2).permute(0, 2, 1) -> down_1
        x2 = ops.conv1d_k3(x1, *p[1])
        x3 = ops.conv1d_k3(x2, *p[2])
        y = ops.conv1d_k3(x3, *p[3])
        y = ops.conv1d_k3(y, *p[4])
        y = ops.conv1d_k3(y, *p[5], src1=x3)                                # cat([x, x3], 1) -> up_1
        y = ops.conv1d_k3(y, *p[6], src1=x2)
        y = ops.conv1d_k3(y, *p[7], src1=x1)
        return ops.conv1d_k3(y, *p[8], dst_nlc=True, act=ops.ACT_SIGMOID)   # predictor -> sigmoid -> permute
