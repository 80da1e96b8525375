"""A/B of the Winograd F(4x4, 3x3) forward kernel (tnv3_conv3x3_wino43_forward) against the production F(2x2, 3x3) kernels
(`variant` -1) on TrackNet's plain-layer shapes that it supports, batch 10: ms per call (eval-mode epilogue: BN affine + ReLU),
effective TFLOP/s at the direct form's count, and the largest difference between the two results relative to the output scale."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops

SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128), (256, 512, 36, 64),
          (512, 512, 36, 64))


def timeit(fn, reps=10):
    for _ in range(3):
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
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * (2.0 / (cin * 9) ** 0.5)
        mean, scale, shift = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        u2, u4 = ops.pack_wino_weights(wt), ops.pack_wino43_weights(wt)
        f2 = lambda: ops.conv3x3_wino(x, u2, cout, mean=mean, scale=scale, shift=shift, relu=True)
        f4 = lambda: ops.conv3x3_wino43(x, u4, cout, mean=mean, scale=scale, shift=shift, relu=True)
        y2, y4 = f2(), f4()
        ref = torch.relu((torch.nn.functional.conv2d(x[:2].double(), wt.double(), padding=1) - mean.double()[None, :, None, None])
                         * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
        mag = ref.abs().max().item()
        row = {"err_f22_vs_fp64": (y2[:2].double() - ref).abs().max().item() / mag, "err_f43_vs_fp64": (y4[:2].double() - ref).abs().max().item() / mag}
        gf = 2.0 * 9 * cin * cout * h * w * 10 / 1e9
        for rep in range(2):
            for name, fn in (("f22", f2), ("f43", f4)):
                ms = timeit(fn)
                row[name] = {"ms": round(ms, 4), "effective_tflops": round(gf / ms, 1)}
        row["speedup"] = round(row["f22"]["ms"] / row["f43"]["ms"], 3)
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino43_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
