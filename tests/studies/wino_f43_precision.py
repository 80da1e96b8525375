#!/usr/bin/env python
"""Study (not a test): what would Winograd F(4x4, 3x3) cost in accuracy on THIS network?

Runs the oracle's TrackNet(27 -> 8).eval() forward at 288 x 512 with every 3x3 convolution replaced by an fp32 emulation of
  - the direct form (torch's fp32 conv2d),
  - F(2x2, 3x3) -- what the HIP kernels compute (16 products per 4 outputs),
  - F(4x4, 3x3) -- 36 products per 16 outputs, 1.78x fewer than F(2x2) -- with Lavin's interpolation points (0, +-1, +-2, inf) and with
    the kernel's (0, +-3/4, +-3/2, inf),
in eval mode (1 x 288 x 512) and in TRAINING mode (2 x 288 x 512: batch-statistics BatchNorm amplifies the rounding), and compares the heat
maps with the fp64 direct forward.  The parity bar of the path is 1e-4 on the heat maps (the HIP F(2x2) kernels
measure 1.6-2.5e-5).  usage: python tests/studies/wino_f43_precision.py [out.json]   (CPU, ~2 minutes; imports oracle/: test tooling)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nets  # noqa: E402

BT = {2: torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
      4: torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], dtype=torch.float64)}
G = {2: torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
     4: torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=torch.float64)}
AT = {2: torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64),
      4: torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)}


def toom_cook(points, m=4, r=3):
    """F(m, r) matrices for the finite points + infinity (fp64): A^T = V_m^T, G = V_r with the Lagrange denominators, B^T = rows of monic
    polynomials prod_{q != p} (x - q)."""
    import numpy as np
    n = m + r - 1
    at = np.zeros((m, n)); g = np.zeros((n, r)); bt = np.zeros((n, n))
    for k, p in enumerate(points):
        at[:, k] = [p ** i for i in range(m)]
        g[k] = [p ** j for j in range(r)]
        g[k] /= np.prod([p - q for q in points if q != p])
        bt[k, :n - 1] = np.poly([q for q in points if q != p])[::-1]
    at[m - 1, n - 1] = 1.0; g[n - 1, r - 1] = 1.0
    bt[n - 1] = np.poly(list(points))[::-1]
    return torch.tensor(bt), torch.tensor(g), torch.tensor(at)


BT[43], G[43], AT[43] = toom_cook([0.0, 0.75, -0.75, 1.5, -1.5])      # the kernel's points (mode 43 below)


def wino_conv(x, w, m):
    """fp32 Winograd F(m x m, 3x3) 'same' convolution: every transform and the channel sum in fp32."""
    n, c, h, wd = x.shape
    bt, g, at = BT[m].float(), G[m].float(), AT[m].float()
    m = 4 if m == 43 else m
    t = m + 2
    xp = F.pad(x, (1, 1 + (-wd) % m, 1, 1 + (-h) % m))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)                        # n, c, th, tw, t, t
    th, tw = tiles.shape[2], tiles.shape[3]
    v = torch.einsum("ij,nchwjk,lk->nchwil", bt, tiles, bt)           # B^T d B
    u = torch.einsum("ij,ocjk,lk->ocil", g, w, g)                     # G g G^T
    mm = torch.einsum("ocil,nchwil->nohwil", u, v)                    # sum over input channels, per (i, l)
    y = torch.einsum("ij,nohwjk,lk->nohwil", at, mm, at)              # A^T M A: n, o, th, tw, m, m
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], th * m, tw * m)
    return y[:, :, :h, :wd].contiguous()


def forward(sd, x, mode, training=False):
    real = F.conv2d

    def conv(inp, weight, bias=None, stride=1, padding=0, *a, **k):
        if weight.shape[-1] == 3 and mode in (2, 4, 43):
            return wino_conv(inp, weight, mode)
        return real(inp, weight, bias, stride, padding, *a, **k)
    F.conv2d = conv
    try:
        with torch.no_grad():
            return nets.tracknet_forward(sd, x, training=training)
    finally:
        F.conv2d = real


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    in_dim, out_dim, h, w = 27, 8, 288, 512
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), 31, calibrated=True)
    out = {"workload": "oracle TrackNet(27, 8), 288 x 512, calibrated synthetic state (the golden generator's), heat maps vs the fp64 direct forward: eval mode "
                       "(1 sample) and training mode (2 samples, batch-statistics BatchNorm)", "parity_bar": 1e-4}
    for training, n in ((False, 1), (True, 2)):
        x = nets.synth_input((n, in_dim, h, w), 77)
        ref = forward({k: v.double() for k, v in sd.items()}, x.double(), 0, training)
        sec = out["training" if training else "eval"] = {}
        for name, mode in (("direct_fp32", 0), ("winograd_F2x2_fp32", 2), ("winograd_F4x4_fp32_points_0_1_2", 4), ("winograd_F4x4_fp32_points_0_3/4_3/2", 43)):
            y = forward(sd, x, mode, training)
            d = (y.double() - ref).abs()
            sec[name] = {"max_abs_err": float(d.max()), "mean_abs_err": float(d.mean()), "p99.99_abs_err": float(d.flatten().kthvalue(int(d.numel() * 0.9999)).values)}
            print("training" if training else "eval", name, sec[name], flush=True)
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
