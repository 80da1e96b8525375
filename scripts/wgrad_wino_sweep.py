"""Direct 3x3 weight-gradient kernel vs the Winograd-form one on the plain TrackNet layer shapes (batch 10)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops


def timeit(fn, reps=6):
    for _ in range(2):
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
    out = {}
    for cin, cout, h, w in ((64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128),
                            (256, 512, 36, 64), (512, 512, 36, 64)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        dz = torch.randn(10, cout, h, w, device=dev) * 0.1
        a = ops.conv3x3_wgrad(x, dz)
        b = ops.conv3x3_wgrad_wino(x, dz)
        err = ((a - b).abs().max() / a.abs().max()).item()
        t_d = timeit(lambda: ops.conv3x3_wgrad(x, dz))
        t_w = timeit(lambda: ops.conv3x3_wgrad_wino(x, dz))
        gf = 2.0 * 9 * cin * cout * h * w * 10 / 1e9
        out[f"{cout},{cin},10,{h},{w}"] = {"direct_ms": round(t_d, 4), "wino_ms": round(t_w, 4), "direct_tflops": round(gf / t_d, 1),
                                          "wino_algorithmic_tflops": round(gf / t_w, 1), "rel_diff": float(f"{err:.2e}")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
