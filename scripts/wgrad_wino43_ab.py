"""A/B of the F(4x4) weight-gradient kernel (variant 8, kernels/wgrad_wino43_mfma.h) against the F(2x2) generations (1: the training
default, 5: the fastest one per call) on TrackNet's plain-layer shapes, batch 10: ms per call (kernel + fold), the distance of the F(4x4)
gradient from the F(2x2) one in units of max|dW| (F(2x2) itself: 1.2e-7 rms / 6e-7 max from fp64 autograd).
usage: wgrad_wino43_ab.py [variant ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops

SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128),
          (256, 512, 36, 64), (512, 512, 36, 64))


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    variants = [int(v) for v in sys.argv[1:]] or [1, 5, 8]
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        dz = torch.randn(10, cout, h, w, device=dev) * 0.1
        row, ref = {}, None
        for rep in range(2):                      # second round: clocks settled, order effects visible
            for v in variants:
                if cin % 64 and v < 5:
                    continue
                dw = ops.conv3x3_wgrad_wino(x, dz, variant=v)
                ms = timeit(lambda: ops.conv3x3_wgrad_wino(x, dz, variant=v))
                row[f"v{v}"] = {"ms": round(ms, 4)}
                if ref is None:
                    ref = dw
                else:
                    d = (dw.double() - ref.double()).abs()
                    row[f"v{v}"]["max_diff_to_first_over_max"] = float(d.max() / ref.double().abs().max())
                    row[f"v{v}"]["rms_diff_to_first_over_max"] = float(d.pow(2).mean().sqrt() / ref.double().abs().max())
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wgrad_wino43_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
