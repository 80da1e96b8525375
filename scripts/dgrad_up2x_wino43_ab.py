"""A/B of the data gradient of the decoder entries' upsampled halves: the one-GEMM F(2x2) kernel (variant 0: 9 multiply-adds per low-res pixel)
vs the 25-of-36 F(4x4) form on the 16x16x4 kernel (variant 2, MODE 2 of kernels/conv3x3_wino43s_mfma.h: 6.25), at TrackNet's three decoder
entries, batch 10: ms per call, executed TFLOP/s as a share of the 157.3 TFLOP/s fp32 MFMA peak, error vs fp64 autograd (batch 2).
  PARTS=custom CUSTOM_CMD="python scripts/dgrad_up2x_wino43_ab.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
    torch.manual_seed(0)
    out = {}
    for c0, cout, hl, wl in SHAPES:
        dz = torch.randn(10, cout, 2 * hl, 2 * wl, device=dev)
        wt = (torch.rand(cout, c0 + c0 // 2, 3, 3, device=dev) - 0.5) * (2.0 / (cout * 9) ** 0.5)
        xl = torch.zeros(2, c0, hl, wl, dtype=torch.float64, requires_grad=True)
        F.conv2d(xl.repeat_interleave(2, 2).repeat_interleave(2, 3), wt[:, :c0].double().cpu(), padding=1).backward(dz[:2].double().cpu())
        ref, mag = xl.grad, xl.grad.abs().max().item()
        row, fns = {}, {}
        for v in (0, 2):
            u = ops.pack_dgrad_up2x_wino_weights(wt, c0, variant=v)
            fns[v] = (lambda u=u, v=v: ops.dgrad_up2x_wino(dz, u, c0, variant=v))
            y = ops.dgrad_up2x_wino(dz[:2].contiguous(), u, c0, variant=v)
            row[f"err_v{v}_vs_fp64"] = (y.double().cpu() - ref).abs().max().item() / mag
        for rep in range(2):
            for v in (0, 2):
                ms = timeit(fns[v])
                gf = 2.0 * PER_LOWRES_PIXEL[v] * c0 * cout * hl * wl * 10 / 1e9
                row[f"v{v}"] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1), "of_mfma_peak": round(gf / ms / 157.3, 3)}
        row["speedup_v2_over_v0"] = round(row["v0"]["ms"] / row["v2"]["ms"], 3)
        out[f"d_up({c0})<-{cout}@{2 * hl}x{2 * wl}"] = row
        print(f"d_up({c0})<-{cout}@{2 * hl}x{2 * wl}", json.dumps(row), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "dgrad_up2x_wino43_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
