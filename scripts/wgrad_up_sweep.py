"""Decoder-entry weight gradient: low-resolution 2x2-window kernels (tnv3_conv3x3_wgrad_up2x) vs the Winograd-form weight
gradient on the materialised upsampled tensor (same multiply-add count: 4/9 vs 16/36)."""
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
    n = 10
    out = {}
    for c0, c1, cout, hl, wl in ((512, 256, 256, 36, 64), (256, 128, 128, 72, 128), (128, 64, 64, 144, 256)):
        x_low = torch.relu(torch.randn(n, c0, hl, wl, device=dev))
        skip = torch.relu(torch.randn(n, c1, 2 * hl, 2 * wl, device=dev))
        dz = torch.randn(n, cout, 2 * hl, 2 * wl, device=dev)

        def via_wino():
            x_up = F.interpolate(x_low, scale_factor=2, mode="nearest")
            return torch.cat([ops.conv3x3_wgrad_wino(x_up, dz), ops.conv3x3_wgrad_wino(skip, dz)], 1)

        a = ops.conv3x3_wgrad_up2x(x_low, skip, dz)
        b = via_wino()
        err = ((a - b).abs().max() / a.abs().max()).item()
        t_a = timeit(lambda: ops.conv3x3_wgrad_up2x(x_low, skip, dz))
        t_old = timeit(lambda: ops.conv3x3_wgrad_up2x(x_low, skip, dz, wino_variant=1))
        t_b = timeit(via_wino)
        t_up = timeit(lambda: F.interpolate(x_low, scale_factor=2, mode="nearest"))
        out[f"{c0}+{c1}->{cout}@{2*hl}x{2*wl}"] = {"up2x_ms": round(t_a, 4), "up2x_2x2_windows_ms": round(t_old, 4), "wino_on_upsampled_ms": round(t_b, 4), "upsample_ms": round(t_up, 4),
                                                  "rel_diff": float(f"{err:.2e}")}
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wgrad_up_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
