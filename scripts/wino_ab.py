"""A/B of the Winograd forward kernel variants on the plain layer shapes of TrackNet (batch 10): ms per launch, executed
TFLOP/s, and bit-equality of the outputs.  usage: wino_ab.py [variant ...]   (default: 2 3)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diaglib

SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128),
          (256, 512, 36, 64), (512, 512, 36, 64))


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
    variants = [int(v) for v in sys.argv[1:]] or [3, 5]
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        sc, sh, mu = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev), torch.rand(cout, device=dev)
        gf = 2.0 * 16 * cin * cout * (h // 2) * (w // 2) * 10 / 1e9
        row, ref = {}, None
        for v in variants:
            if v in (6,) or 100 <= v < 120:
                if cout % 128 or cin <= 8:
                    continue
            if v in (7,) or 120 <= v < 130:
                if cout % 64 or cin <= 8:
                    continue
            u = ops.pack_wino_weights(wt, variant=v)      # the panel layout follows the kernel
            if v >= 10 or v in (0, 2, 4):  # experimental arms and the rejected generations live in libtnv3_diag.so (raw convolution, no affine)
                y = torch.empty(10, cout, h, w, device=dev)
                run = lambda: diaglib.conv3x3_wino_forward(x, u, y, v)       # noqa: E731
                run()
            else:
                y = ops.conv3x3_wino(x, u, cout, variant=v)
                run = lambda: ops.conv3x3_wino(x, u, cout, variant=v)        # noqa: E731
            ms = timeit(run)
            row[f"v{v}"] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1)}
            if ref is None:
                ref = y
            else:
                row[f"v{v}"]["bit_equal_to_first"] = bool(torch.equal(ref, y))
                row[f"v{v}"]["max_abs_diff"] = float((ref - y).abs().max())
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
