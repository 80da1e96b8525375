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
            self.out = inpaintnet_forward(net, self.x, self.m)

    def replay(self):
        self.graph.replay()
        return self.out

    def __call__(self, x, m):
        self.x.copy_(x)
        self.m.copy_(m)
        return self.replay()
