"""A/B of the upsampled half's two Winograd forms (tnv3_conv_up2x_wino_forward variant 0 = 9 of the 16 F(2x2) GEMMs, variant 2 = 25 of the 36
F(4x4) products on the 16x16x4 kernel) at the three decoder entries of TrackNet, batch 10: ms per call, executed TFLOP/s as a share of the
157.3 TFLOP/s fp32 MFMA peak, error vs fp64 torch on the materialised upsampled tensor.
  PARTS=custom CUSTOM_CMD="python scripts/up2x_wino43_ab.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops

SHAPES = ((512, 256, 36, 64), (256, 128, 72, 128), (128, 64, 144, 256))      # (c0, cout, h_low, w_low)
PER_LOWRES_PIXEL = {0: 9.0, 2: 6.25}


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
    for c0, cout, hl, wl in SHAPES:
        xl = torch.relu(torch.randn(10, c0, hl, wl, device=dev))
        wt = (torch.rand(cout, c0 + c0 // 2, 3, 3, device=dev) - 0.5) * (2.0 / (c0 * 9) ** 0.5)
        up = xl[:2].repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = torch.nn.functional.conv2d(up.double(), wt[:, :c0].double(), padding=1)
        mag = ref.abs().max().item()
        row, fns = {}, {}
        for v in (0, 2):
            u = ops.pack_up2x_wino_weights(wt, c0, variant=v)
            fns[v] = (lambda u=u, v=v: ops.conv_up2x_wino(xl, u, cout, variant=v))
            y = fns[v]()
            row[f"err_v{v}_vs_fp64"] = (y[:2].double() - ref).abs().max().item() / mag
        for rep in range(2):
            for v in (0, 2):
                ms = timeit(fns[v])
                gf = 2.0 * PER_LOWRES_PIXEL[v] * c0 * cout * hl * wl * 10 / 1e9
                row[f"v{v}"] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1), "of_mfma_peak": round(gf / ms / 157.3, 3)}
        row["speedup_v2_over_v0"] = round(row["v0"]["ms"] / row["v2"]["ms"], 3)
        out[f"up({c0})->{cout}@{2 * hl}x{2 * wl}"] = row
        print(f"up({c0})->{cout}@{2 * hl}x{2 * wl}", json.dumps(row), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "up2x_wino43_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
