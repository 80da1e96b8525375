"""A/B of the Winograd-form weight-gradient kernels (variant 0: one wave per SIMD, phases alternate; 1: two waves per SIMD,
wave groups half a period apart; 2 / 3: 16-byte operand reads, two / three raw stages; >= 100: timing twins through libtnv3_diag.so)
on TrackNet's plain-layer shapes, batch 10: ms per call (kernel + fold), executed TFLOP/s, bit-equality.
usage: wgrad_wino_ab.py [variant ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def wgrad(x, dz, v):
    if v >= 100 or v in (0, 3, 4, 6, 7):      # timing twins and the rejected generations: libtnv3_diag.so
        import diaglib
        return diaglib.conv3x3_wgrad_wino(x, dz, v)
    return ops.conv3x3_wgrad_wino(x, dz, variant=v)

SHAPES = ((64, 64, 288, 512), (64, 128, 144, 256), (128, 128, 144, 256), (128, 256, 72, 128), (256, 256, 72, 128), (256, 512, 36, 64),
          (512, 512, 36, 64))


def timeit(fn, reps=8):
    for _ in range(2):
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
    variants = [int(v) for v in sys.argv[1:]] or [1, 5]
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        dz = torch.randn(10, cout, h, w, device=dev) * 0.1
        gf = 2.0 * 16 * cin * cout * (h // 2) * (w // 2) * 10 / 1e9
        row, ref = {}, None
        for rep in range(2):                      # second round: clocks settled, order effects visible
            for v in variants:
                dw = wgrad(x, dz, v)
                ms = timeit(lambda: wgrad(x, dz, v))
                row[f"v{v}"] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1)}
                if ref is None:
                    ref = dw
                else:
                    row[f"v{v}"]["bit_equal_to_first"] = bool(torch.equal(ref, dw))
        out[f"{cin}->{cout}@{h}x{w}"] = row
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(row), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wgrad_wino_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
