"""A/B of the decoder entries' weight gradient, upsampled half in the 9-GEMM F(2x2) form (up_variant 1) vs the 25-of-36 F(4x4) form
(up_variant 2, kernels/wgrad_up2x_wino43_mfma.h), at TrackNet's three decoder entries, batch 10: ms per call of the whole entry (upsampled
half + skip half + folds + join) and the gradient's distance from fp64 autograd on the materialised upsampled tensor (batch 2).
  PARTS=custom CUSTOM_CMD="python scripts/wgrad_up2x_wino43_ab.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tracknetv3_amd import ops


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    out = {}
    for c0, c1, cout, hl, wl in ((512, 256, 256, 36, 64), (256, 128, 128, 72, 128), (128, 64, 64, 144, 256)):
        x_low = torch.relu(torch.randn(10, c0, hl, wl, device=dev))
        skip = torch.relu(torch.randn(10, c1, 2 * hl, 2 * wl, device=dev))
        dz = torch.randn(10, cout, 2 * hl, 2 * wl, device=dev)
        xu = F.interpolate(x_low[:2], scale_factor=2, mode="nearest").double().cpu()
        wd = torch.zeros((cout, c0, 3, 3), dtype=torch.float64, requires_grad=True)
        F.conv2d(xu, wd, padding=1).backward(dz[:2].double().cpu())
        ref = wd.grad
        row = {}
        for v in (1, 2):
            g = ops.conv3x3_wgrad_up2x(x_low[:2].contiguous(), skip[:2].contiguous(), dz[:2].contiguous(), up_variant=v)[:, :c0].double().cpu()
            row[f"up_variant_{v}_rel_err_vs_fp64"] = ((g - ref).abs().max() / ref.abs().max()).item()
        for rep in range(2):
            for v in (1, 2):
                row[f"up_variant_{v}_ms"] = round(timeit(lambda: ops.conv3x3_wgrad_up2x(x_low, skip, dz, up_variant=v)), 4)
        row["speedup"] = round(row["up_variant_1_ms"] / row["up_variant_2_ms"], 3)
        out[f"up({c0})+{c1}->{cout}@{2 * hl}x{2 * wl}"] = row
        print(f"up({c0})+{c1}->{cout}@{2 * hl}x{2 * wl}", json.dumps(row), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "wgrad_up2x_wino43_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
