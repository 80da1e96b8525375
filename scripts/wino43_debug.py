"""(debug aid) F(4x4) kernel's statistics epilogue on the GPU: where do wrong elements of z sit, are they unwritten or overwritten?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.nn.functional as F
from tracknetv3_amd import ops
from test_emu_kernels import T
d = torch.device("cuda", 0)
_empty = torch.empty


def nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point():
        t.fill_(float("nan"))
    return t


def rng(v):
    v = sorted(set(v.tolist()))
    return v if len(v) <= 12 else (v[:6], "..", v[-6:], len(v))


for case in [(1, 16, 64, 8, 64), (2, 20, 128, 16, 64), (2, 64, 64, 64, 128)]:
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 491)).to(d), T((cout, cin, 3, 3), 492, -0.3, 0.3).to(d)
    ref = F.conv2d(x.double(), wt.double(), padding=1); mag = ref.abs().max().item()
    s1r, s2r = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
    u = ops.pack_wino43_weights(wt)
    for it in range(8):
        ops.torch.empty = nan_empty
        try:
            z, st = ops.conv3x3_wino43_stats(x, u, cout)
        finally:
            ops.torch.empty = _empty
        torch.cuda.synchronize()
        zd = z.double()
        bad = ~((zd - ref).abs() <= 1e-4 * mag)
        nb = int(bad.sum())
        s1, s2 = st[:, :, 0].sum(1), st[:, :, 1].sum(1)
        e1 = ((s1 - s1r).abs().max() / s1r.abs().max()).item(); e2 = ((s2 - s2r).abs().max() / s2r.abs().max()).item()
        msg = f"{case} run {it}: bad z {nb} (nan {int(torch.isnan(z).sum())}), stats vs ref {e1:.2e} {e2:.2e}, stats nan {int(torch.isnan(st).sum())}"
        if nb:
            idx = bad.nonzero()
            msg += f"\n   n {rng(idx[:, 0])} c {rng(idx[:, 1])} h {rng(idx[:, 2])} w {rng(idx[:, 3])}\n   sample {[(tuple(i.tolist()), float(z[tuple(i.tolist())]), float(ref[tuple(i.tolist())])) for i in idx[:6]]}"
        print(msg, flush=True)
