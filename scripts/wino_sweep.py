"""Direct 3x3 MFMA kernel vs the fused Winograd F(2x2,3x3) kernel on the plain TrackNet layer shapes (batch 10)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops, tuning


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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out = {}
    for cin, cout, h, w in ((64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128),
                            (256, 512, 36, 64), (512, 512, 36, 64)):
        x = torch.relu(torch.randn(n, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        wp, u = ops.pack_conv3x3_weights(wt), ops.pack_wino_weights(wt)
        sc, sh, mu = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev), torch.rand(cout, device=dev)
        cfg = tuning.conv_config(cout, cin, n, h, w)
        a = ops.conv3x3(x, wp, cout, mean=mu, scale=sc, shift=sh, relu=True, cfg=cfg)
        b = ops.conv3x3_wino(x, u, cout, mean=mu, scale=sc, shift=sh, relu=True)
        err = ((a - b).abs().max() / a.abs().max()).item()
        t_d = timeit(lambda: ops.conv3x3(x, wp, cout, mean=mu, scale=sc, shift=sh, relu=True, cfg=cfg))
        t_w = timeit(lambda: ops.conv3x3_wino(x, u, cout, mean=mu, scale=sc, shift=sh, relu=True))
        other = int(os.environ.get("WINO_OTHER", "0"))               # the Winograd kernel to compare with (per-call variant)
        uo = u
        c = ops.conv3x3_wino(x, uo, cout, mean=mu, scale=sc, shift=sh, relu=True, variant=other)
        t_o = timeit(lambda: ops.conv3x3_wino(x, uo, cout, mean=mu, scale=sc, shift=sh, relu=True, variant=other))
        err_o = ((a - c).abs().max() / a.abs().max()).item()
        gf = 2.0 * 9 * cin * cout * h * w * n / 1e9
        out[f"{cout},{cin},{n},{h},{w}"] = {"direct_ms": round(t_d, 4), "wino_ms": round(t_w, 4), "direct_tflops": round(gf / t_d, 1),
                                           "wino_algorithmic_tflops": round(gf / t_w, 1), "wino_executed_tflops": round(gf * 16 / 36 / t_w, 1),
                                           "rel_diff": float(f"{err:.2e}"),
                                           "variant%d_ms" % other: round(t_o, 4), "variant%d_rel_diff" % other: float(f"{err_o:.2e}")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
