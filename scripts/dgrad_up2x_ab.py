"""Low-resolution data gradient of the three decoder-entry layers (batch 10): the 4x4 stride-2 correlation (dgrad_up2x, 16
multiply-adds per low-res pixel and channel pair) vs the one-GEMM Winograd form (dgrad_up2x_wino, K = 9 * Cout)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops


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
    n = 10
    out = {}
    for c0, c1, cout, hl, wl in ((512, 256, 256, 36, 64), (256, 128, 128, 72, 128), (128, 64, 64, 144, 256)):
        dz = torch.randn(n, cout, 2 * hl, 2 * wl, device=dev) * 0.1
        w = (torch.rand(cout, c0 + c1, 3, 3, device=dev) - 0.5) * 0.1
        g, u = ops.pack_dgrad_up2x_weights(w, c0), ops.pack_dgrad_up2x_wino_weights(w, c0)
        a, b = ops.dgrad_up2x(dz, g, c0), ops.dgrad_up2x_wino(dz, u, c0)
        row = {"rel_diff": float(f"{((a - b).abs().max() / a.abs().max()).item():.2e}")}
        for rep in range(2):
            t_a = timeit(lambda: ops.dgrad_up2x(dz, g, c0))
            t_b = timeit(lambda: ops.dgrad_up2x_wino(dz, u, c0))
        lowpix = n * hl * wl
        row["stride2_4x4"] = {"ms": round(t_a, 4), "executed_tflops": round(2.0 * 16 * c0 * cout * lowpix / t_a / 1e9, 1)}
        row["one_gemm_k9"] = {"ms": round(t_b, 4), "executed_tflops": round(2.0 * 9 * c0 * cout * lowpix / t_b / 1e9, 1)}
        b1 = ops.dgrad_up2x_wino(dz, u, c0, variant=1)                        # round 2's group order (younger waves' MFMAs first)
        t_c = min(timeit(lambda: ops.dgrad_up2x_wino(dz, u, c0, variant=1)) for _ in range(2))
        t_b2 = min(timeit(lambda: ops.dgrad_up2x_wino(dz, u, c0)) for _ in range(2))
        row["one_gemm_k9_younger_first"] = {"ms": round(t_c, 4), "bit_equal": bool(torch.equal(b, b1)), "older_first_ms_again": round(t_b2, 4)}
        out[f"{c0}<-{cout}@{hl}x{wl}"] = row
        print(f"{c0}<-{cout}@{hl}x{wl}", json.dumps(row), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dgrad_up2x_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
