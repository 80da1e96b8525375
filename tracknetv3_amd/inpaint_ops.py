"""InpaintNet.forward (model.py:113-129) as nine conv1d launches of libtnv3_hip.so; concats are two-pointer reads."""
import torch

from . import ops


def inpaintnet_forward(net, x, m):
    if net.training and torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
        from . import autograd_ops
        return autograd_ops.inpaintnet_forward_train(net, x, m)
    with torch.no_grad():
        x = x.contiguous().float()
        m = m.contiguous().to(torch.float32)
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
