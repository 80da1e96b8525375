"""A/B of the two Winograd F(4x4, 3x3) kernels (tnv3_conv3x3_wino43_forward variant 0 = 16x16x4 MFMAs / one wave per block / write-out in
registers, variant 1 = 32x32x2 / four waves per block) on TrackNet's plain-layer shapes, batch 10: ms per call (eval-mode epilogue: BN
affine + ReLU), executed TFLOP/s (36 products per 4x4 tile) as a share of the 157.3 TFLOP/s fp32 MFMA peak, error vs fp64 torch.
  PARTS=custom CUSTOM_CMD="python scripts/wino43_variant_ab.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops

SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128), (256, 512, 36, 64),
          (512, 512, 36, 64), (512, 256, 72, 128), (256, 128, 144, 256), (128, 64, 288, 512))
VARIANTS = tuple(int(v) for v in sys.argv[1:]) or (0, 2, 1)


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
        ref = torch.relu((torch.nn.functional.conv2d(x[:2].double(), wt.double(), padding=1) - mean.double()[None, :, None, None])
                         * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
        mag = ref.abs().max().item()
        fns, row = {}, {}
        for v in VARIANTS:
            u = ops.pack_wino43_weights(wt, variant=v)
            fns[v] = (lambda u=u, v=v: ops.conv3x3_wino43(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, variant=v))
            y = fns[v]()
            row[f"err_v{v}_vs_fp64"] = (y[:2].double() - ref).abs().max().item() / mag
            y1 = fns[v]()
            row[f"repeatable_v{v}"] = bool(torch.equal(y, y1))
        gf_exec = 2.0 * 9 * cin * cout * h * w * 10 / 1e9 / 4.0
        for rep in range(2):
            for v in VARIANTS:
                ms = timeit(fns[v])
                row[f"v{v}"] = {"ms": round(ms, 4), "executed_tflops": round(gf_exec / ms, 1), "of_mfma_peak": round(gf_exec / ms / 157.3, 3)}
        if 1 in VARIANTS:
            for v in VARIANTS:
                if v != 1:
                    row[f"speedup_v{v}_over_v1"] = round(row["v1"]["ms"] / row[f"v{v}"]["ms"], 3)
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "wino43_variant_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
