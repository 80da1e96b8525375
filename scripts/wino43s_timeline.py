"""Cycle-level totals of the 16x16x4 Winograd F(4x4, 3x3) kernel (conv3x3_wino43s_kernel) and of its timing twins: per layer shape at batch 10,
cycles per step (one chunk of one tile: 4608 cycles of MFMAs per SIMD), cycles per tile in the write-out, the prologue, and the wall time of
the launch -- for the product kernel's work (ring 6 / 9 / 18 quads of filter operand) and with parts of a step removed (WRONG results):
mask bit 0 no raw DMA, 1 no patch transform, 2 no A loads, 3 no B reads, 4 no MFMAs, 5 no output stores.
  PARTS=custom CUSTOM_CMD="python scripts/wino43s_timeline.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tracknetv3_amd import ops
import diaglib

TWINS = ((6, 0), (9, 0), (18, 0), (6, 1), (6, 2), (6, 3), (6, 4), (6, 7), (6, 15), (6, 16), (6, 32), (18, 3), (18, 7))
SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (128, 128, 144, 256), (256, 256, 72, 128), (512, 512, 36, 64))


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        u = ops.pack_wino43_weights(wt, variant=0)
        y = torch.empty(10, cout, h, w, device=dev)
        tl = torch.zeros(32, dtype=torch.int64, device=dev)
        ref = ops.conv3x3_wino43(x, u, cout, variant=0)
        rows = {}
        for ring, mask in TWINS:
            for _ in range(2):
                diaglib.conv3x3_wino43s_timeline(x, u, y, tl, ring, mask)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                diaglib.conv3x3_wino43s_timeline(x, u, y, tl, ring, mask)
            e1.record()
            torch.cuda.synchronize()
            raw = tl.cpu().reshape(4, 8).double()
            steps, tiles = raw[:, 3].mean().item(), raw[:, 4].mean().item()
            d = {"ms": round(e0.elapsed_time(e1) / 5, 4), "cycles_per_step": round(raw[:, 1].mean().item() / max(steps, 1), 1),
                 "write_out_cycles_per_tile": round(raw[:, 2].mean().item() / max(tiles, 1), 1), "prologue_cycles": round(raw[:, 0].mean().item(), 1),
                 "steps": steps, "tiles": tiles}
            if mask == 0:
                d["equal_to_product_kernel"] = bool(torch.equal(y, ref))
            rows[f"ring{ring}_mask{mask}"] = d
            print(f"{cin}->{cout}@{h}x{w} ring {ring} mask {mask}", json.dumps(d), flush=True)
        out[f"{cin}->{cout}@{h}x{w}"] = rows
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "wino43s_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
