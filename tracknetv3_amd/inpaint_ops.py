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


class GraphedInpaintNet:
    """The eval forward of an InpaintNet for ONE batch size, captured in a HIP graph: at the reference's batch of 32
    sequences (README.md:162) the nine launches are launch-latency bound, and a replay costs one submission.  Inputs are
    copied into the static buffers `x` / `m` (or written there directly by the caller), `replay()` returns the static
    output.  The graph reads the parameters in place, so in-place weight updates (load_state_dict, an optimiser step) are
    picked up; moving the module to another device or dtype needs a new capture."""

    def __init__(self, net, n, device=None, seq_len=16):
        if net.training:
            raise ValueError("GraphedInpaintNet captures the eval forward: call net.eval() first")
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != "cuda":
            raise ValueError("HIP graphs need a GPU")
        self.net = net
        self.x = torch.zeros((int(n), int(seq_len), 2), dtype=torch.float32, device=dev)
        self.m = torch.zeros((int(n), int(seq_len), 1), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(dev)                      # warm-up off the default stream: first launches load code objects
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                inpaintnet_forward(net, self.x, self.m)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if use_fused(int(n), int(seq_len)):
                # the fused kernel reads a PACKED copy of the parameters: the pack launch is part of the graph, so that a replay
                # follows in-place weight updates exactly like the eager forward (two kernels per replay)
                params = net.conv_params()
                packed = ops.inpaintnet_pack([w.detach() for w, _ in params], [b.detach() for _, b in params])
                self.out = ops.inpaintnet_fused(self.x, self.m, packed)
            else:
                self.out = inpaintnet_forward(net, self.x, self.m)

    def replay(self):
        self.graph.replay()
        return self.out

    def __call__(self, x, m):
        self.x.copy_(x)
        self.m.copy_(m)
        return self.replay()
