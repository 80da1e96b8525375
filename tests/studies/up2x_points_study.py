#!/usr/bin/env python
"""Study (not a test): which interpolation points for the 25-of-36 form of the decoder entries' upsampled halves?

A 3x3 convolution over nn.Upsample(2)(l) in Winograd F(4x4, 3x3) form loses the products of the point -1 whatever the other points are
(the upsampled signal's polynomial is (1 + x) l(x^2)), so the set is (0, +-1, +-b, inf) with b free.  This emulates the F(4x4) form in fp32
(every transform and the channel sum in fp32, as tests/studies/wino_f43_precision.py does for the whole network) on upsampled post-ReLU
inputs for a range of b, next to the direct fp32 form, F(2x2) and the plain layers' set (0, +-3/4, +-3/2, inf) -- which is NOT available
here (it has no point -1) and marks what a 36-product form reaches.  Reports max and rms error vs fp64, relative to the output scale, and
the rms by position inside the 4x4 tile (the error sits in output row / column 3: the point at infinity and b^3).
usage: python tests/studies/up2x_points_study.py [out.json]      (CPU, ~1 minute)"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wino_f43_precision as S  # noqa: E402


def main():
    torch.manual_seed(0)
    out = {}
    for K, cout in ((512, 64), (256, 64), (128, 64)):
        w = (torch.rand(cout, K, 3, 3) - 0.5) * (2.0 / (K * 9) ** 0.5)
        xl = torch.relu(torch.randn(2, K, 12, 16))
        xu = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = F.conv2d(xu.double(), w.double(), padding=1)
        mag = ref.abs().max().item()

        def err(y):
            d = (y.double() - ref).abs() / mag
            e = d.reshape(2, cout, 6, 4, 8, 4)
            return {"max": d.max().item(), "rms": d.pow(2).mean().sqrt().item(),
                    "rms_tile_corner_00_vs_33": [e[:, :, :, 0, :, 0].pow(2).mean().sqrt().item(), e[:, :, :, 3, :, 3].pow(2).mean().sqrt().item()]}
        row = {"direct_fp32": err(F.conv2d(xu, w, padding=1)), "F2x2": err(S.wino_conv(xu, w, 2)),
               "F4x4_plain_points_0_3/4_3/2 (no point -1: 36 products)": err(S.wino_conv(xu, w, 43))}
        keep = (S.BT[4], S.G[4], S.AT[4])
        for b2 in (4.0, 3.5, 3.0625, 2.75, 2.5, 2.25, 2.0, 1.5, 0.5, 0.25):
            b = b2 ** 0.5
            S.BT[4], S.G[4], S.AT[4] = S.toom_cook([0.0, 1.0, -1.0, b, -b])
            row[f"F4x4_points_0_1_b^2={b2}"] = err(S.wino_conv(xu, w, 4))
        S.BT[4], S.G[4], S.AT[4] = keep
        out[f"up({K})->{cout}"] = row
        for k, v in row.items():
            print(f"K={K:3d} {k:58s} max {v['max']:.2e} rms {v['rms']:.2e} corners {v['rms_tile_corner_00_vs_33'][0]:.1e} / {v['rms_tile_corner_00_vs_33'][1]:.1e}", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
