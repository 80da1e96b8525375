#!/usr/bin/env python
"""Study (not a test): what would a Winograd F(4x4, 3x3) WEIGHT gradient cost in accuracy?

dW = G^T [ sum over images and tiles of (A dY A^T) * (B^T d B) ] G -- 36 products per 4x4 output tile instead of F(2x2)'s 64 -- evaluated
in fp32 (transforms and the tile sum; torch's blocked summation, so the absolute figures are optimistic for a sequential MFMA chain, the
RATIOS between the forms are what counts) against fp64 autograd, on two layer shapes of the network at batch 10, with Lavin's
interpolation points and with the forward kernel's (0, +-3/4, +-3/2, inf).  usage: python tests/studies/wino_f43_wgrad_precision.py [out.json]
(CPU, ~1 minute)"""
import json
import sys
import numpy as np
import torch
import torch.nn.functional as F
torch.manual_seed(0); torch.set_num_threads(64)
def toom_cook(points, m, r=3):
    n = m + r - 1
    at = np.zeros((m, n)); g = np.zeros((n, r)); bt = np.zeros((n, n))
    for k, p in enumerate(points):
        at[:, k] = [p ** i for i in range(m)]
        g[k] = [p ** j for j in range(r)]
        g[k] /= np.prod([p - q for q in points if q != p])
        bt[k, :n - 1] = np.poly([q for q in points if q != p])[::-1]
    at[m - 1, n - 1] = 1.0; g[n - 1, r - 1] = 1.0
    bt[n - 1] = np.poly(list(points))[::-1]
    return at, g, bt
def wgrad_wino(x, dy, pts, m, dtype):
    """dW[o][c] = G^T [ sum_tiles (A dY A^T) * (B^T d B) ] G, transforms and the tile sum in `dtype`."""
    at, g, bt = (torch.tensor(a, dtype=dtype) for a in toom_cook(pts, m))
    n, c, h, w = x.shape; t = m + 2
    xp = F.pad(x.to(dtype), (1, 1, 1, 1))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)                       # n c th tw t t
    v = torch.einsum("ij,nchwjk,lk->nchwil", bt, tiles, bt)
    dyt = dy.to(dtype).unfold(2, m, m).unfold(3, m, m)               # n o th tw m m
    yh = torch.einsum("ji,nohwjk,kl->nohwil", at, dyt, at)           # A dY A^T  (A = at^T)
    du = torch.einsum("nohwil,nchwil->ocil", yh, v)                  # sum over images and tiles
    return torch.einsum("ji,ocjk,kl->ocil", g, du, g)                # G^T dU G
out = {}
for (n, c, o, h, w) in [(10, 64, 64, 64, 128), (10, 128, 128, 32, 64)]:
    sec = out[f"{c}->{o}@{h}x{w} batch {n}"] = {}
    x = torch.relu(torch.randn(n, c, h, w)); dy = torch.randn(n, o, h, w) * 0.01
    xd = x.double(); wd = torch.zeros(o, c, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wd, padding=1).backward(dy.double()); ref = wd.grad; mag = ref.abs().max().item()
    chk = wgrad_wino(x, dy, [0, 1, -1], 2, torch.float64); assert (chk - ref).abs().max().item() < 1e-9 * mag
    ws = torch.zeros(o, c, 3, 3, requires_grad=True); F.conv2d(x, ws, padding=1).backward(dy); d = (ws.grad.double() - ref).abs()
    sec["direct_fp32"] = {"max": d.max().item() / mag, "rms": d.pow(2).mean().sqrt().item() / mag}
    for name, pts, m in (("F(2x2)", [0, 1, -1], 2), ("F(4x4) Lavin", [0, 1, -1, 2, -2], 4), ("F(4x4) x3/4", [0, .75, -.75, 1.5, -1.5], 4)):
        chk = wgrad_wino(x, dy, pts, m, torch.float64); assert (chk - ref).abs().max().item() < 1e-8 * mag, name
        d = (wgrad_wino(x, dy, pts, m, torch.float32).double() - ref).abs()
        sec[name] = {"max": d.max().item() / mag, "rms": d.pow(2).mean().sqrt().item() / mag}
    print(json.dumps({k: sec}), flush=True) if (k := f"{c}->{o}@{h}x{w}") else None
out["unit"] = "error vs fp64 autograd relative to max |dW|"
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
