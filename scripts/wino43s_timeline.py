"""Cycle-level totals of the 16x16x4 Winograd F(4x4, 3x3) kernel (conv3x3_wino43s_kernel) and of its timing twins: per layer shape at batch 10,
cycles per step (one chunk of one tile: 4608 cycles of MFMAs per SIMD), cycles per tile in the write-out, the prologue, and the wall time of
the launch -- for the product kernel's work (geometry 64 x 2 rows / 128 x 1 row, step schedules grow / ts) and with parts of a step removed (WRONG results):
mask bit 0 no raw DMA, 1 no patch transform, 2 no A loads, 3 no B reads, 4 no MFMAs, 5 no output stores.
  PARTS=custom CUSTOM_CMD="python scripts/wino43s_timeline.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tracknetv3_amd import ops
import diaglib

# (geometry cbw, grow, ts, mask)
TWINS = ((4, 10, 10, 0), (4, 5, 3, 0), (4, 13, 12, 0), (4, 8, 8, 0), (4, 12, 4, 0), (4, 10, 10, 1), (4, 10, 10, 2), (4, 10, 10, 3), (4, 10, 10, 4), (4, 10, 10, 15),
         (4, 10, 10, 16), (8, 10, 10, 0), (8, 5, 4, 0), (8, 13, 12, 0), (8, 10, 28, 0), (8, 16, 40, 0), (8, 10, 10, 1), (8, 10, 10, 2), (8, 10, 10, 3), (8, 10, 10, 4),
         (8, 10, 10, 15), (8, 10, 10, 16))
SHAPES = ((27, 64, 288, 512), (64, 64, 288, 512), (128, 128, 144, 256), (256, 256, 72, 128), (512, 512, 36, 64))


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for cin, cout, h, w in SHAPES:
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        u = ops.pack_wino43_weights(wt, variant=0)
        y = torch.empty(10, cout, h, w, device=dev)
        tl = torch.zeros(64, dtype=torch.int64, device=dev)
        ref = ops.conv3x3_wino43(x, u, cout, variant=0)
        rows = {}
        for cbw, grow, ts, mask in TWINS:
            if cbw == 8 and cout % 128:
                continue
            for _ in range(2):
                diaglib.conv3x3_wino43s_timeline(x, u, y, tl, cbw, grow, ts, mask)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                diaglib.conv3x3_wino43s_timeline(x, u, y, tl, cbw, grow, ts, mask)
            e1.record()
            torch.cuda.synchronize()
            raw = tl.cpu().reshape(8, 8).double()
            steps, tiles = raw[:, 3].mean().item(), raw[:, 4].mean().item()
            d = {"ms": round(e0.elapsed_time(e1) / 5, 4), "cycles_per_step": round(raw[:, 1].mean().item() / max(steps, 1), 1),
                 "write_out_cycles_per_tile": round(raw[:, 2].mean().item() / max(tiles, 1), 1), "prologue_cycles": round(raw[:, 0].mean().item(), 1),
                 "steps": steps, "tiles": tiles}
            if mask == 0:
                d["max_diff_vs_product_kernel"] = (y - ref).abs().max().item()
            rows[f"cbw{cbw}_grow{grow}_ts{ts}_mask{mask}"] = d
            print(f"{cin}->{cout}@{h}x{w} cbw {cbw} grow {grow} ts {ts} mask {mask}", json.dumps(d), flush=True)
        out[f"{cin}->{cout}@{h}x{w}"] = rows
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "wino43s_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
