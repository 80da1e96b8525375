"""Per-tile fixed cost of the Winograd forward kernel: the diag twins 83 (variant 3) and 85 (variant 5, persistent) stamp
s_memtime around the phases OUTSIDE the chunk loop of one mid-grid workgroup -- wait for the first DMAs, first patch transform,
chunk loop, next-tile set-up + DMA issue, output transform + exchange, affine + stores -- and write the totals behind the N-th
image of the output.  Prints cycles per tile and phase, per wave group.  usage: wino_fixed_cost.py [variant ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from tracknetv3_amd import ops
import diaglib

NAMES = ["wait_first_dma", "first_transform", "chunk_loop", "next_tile_setup_and_dma_issue", "out_transform_exchange", "affine_and_stores"]


def main():
    dev = torch.device("cuda", 0)
    variants = [int(v) for v in (sys.argv[1:] or ["83", "85"])]
    out = {}
    for cin, cout, h, w in ((27, 64, 288, 512), (64, 64, 288, 512), (128, 128, 144, 256), (256, 256, 72, 128), (512, 512, 36, 64)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        for v in variants:
            u = ops.pack_wino_weights(wt, variant=3)
            y = torch.zeros(11, cout, h, w, device=dev)          # image 10 receives the phase totals
            for _ in range(3):
                diaglib.conv3x3_wino_forward(x, u, y, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                diaglib.conv3x3_wino_forward(x, u, y, v)
            e1.record()
            torch.cuda.synchronize()
            raw = y[10].reshape(-1)[: 8 * 8 * 2].cpu().numpy().view(np.uint64).reshape(8, 8).astype(np.int64)
            items = int(raw[0, 6])
            d = {"ms_per_launch": round(e0.elapsed_time(e1) / 10, 4), "tiles_of_this_workgroup": items,
                 "workgroup_cycles_per_tile": round(float(raw[:, 7].mean()) / max(items, 1), 1)}
            for name, waves in (("grp0", slice(0, 4)), ("grp1", slice(4, 8))):
                d[name] = {n: round(float(raw[waves, i].mean()) / max(items, 1), 1) for i, n in enumerate(NAMES)}
            chunks = (cin + 7) // 8
            d["chunks"] = chunks
            d["fixed_cycles_per_tile"] = round(d["workgroup_cycles_per_tile"] - d["grp0"]["chunk_loop"], 1)
            out[f"{cin}->{cout}@{h}x{w},v{v}"] = d
            print(f"{cin}->{cout}@{h}x{w} v{v}", json.dumps(d), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino_fixed_cost.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
