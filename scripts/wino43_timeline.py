"""Cycle-level phase totals of the Winograd F(4x4, 3x3) kernel: its timeline twin (libtnv3_diag.so, tnv3_diag_conv3x3_wino43_timeline)
stamps s_memtime at the phase boundaries of every tile of one mid-grid workgroup; this prints, per layer shape at batch 10, the
cycles per chunk inside the chunk loop and the cycles per tile outside it (fill of the first chunk, next tile's offsets + raw issue,
write-out, A issue) -- the "~10 us per tile" that DESIGN 8 derives from layer pairs, measured directly.
  PARTS=custom CUSTOM_CMD="python scripts/wino43_timeline.py" bash scripts/gpu_session.sh"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tracknetv3_amd import ops
import diaglib

NAMES = ["tile_fill", "chunk_loop", "next_offsets_raw_issue", "write_out", "a_issue_tail"]


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for cin, cout, h, w in ((27, 64, 288, 512), (64, 64, 288, 512), (128, 128, 144, 256), (256, 256, 72, 128), (512, 512, 36, 64)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        u = ops.pack_wino43_weights(wt, variant=1)
        y = torch.empty(10, cout, h, w, device=dev)
        tl = torch.zeros(64, dtype=torch.int64, device=dev)
        for _ in range(3):
            diaglib.conv3x3_wino43_timeline(x, u, y, tl)
        torch.cuda.synchronize()
        ref = ops.conv3x3_wino43(x, u, cout, variant=1)
        raw = tl.cpu().reshape(8, 8).double()
        chunks, tiles = raw[:, 5].mean().item(), raw[:, 6].mean().item()
        d = {"tiles_walked": tiles, "chunks_per_tile": chunks / max(tiles, 1), "results_equal_the_product_kernel": bool(torch.equal(y, ref)),
             "cycles_per_chunk_in_loop": round(raw[:, 1].mean().item() / max(chunks, 1), 1)}
        for i, nm in enumerate(NAMES):
            if i != 1:
                d[f"cycles_per_tile_{nm}"] = round(raw[:, i].mean().item() / max(tiles, 1), 1)
        d["cycles_per_tile_outside_the_loop"] = round(sum(raw[:, i].mean().item() for i in (0, 2, 3, 4)) / max(tiles, 1), 1)
        # what the write-out is made of: twins 2 (output stores dropped), 3 (no LDS exchange), 4 (neither) -- wrong results by design
        for variant, nm in ((2, "no_stores"), (3, "no_lds_exchange"), (4, "neither")):
            for _ in range(3):
                diaglib.conv3x3_wino43_timeline(x, u, y, tl, variant)
            torch.cuda.synchronize()
            r2 = tl.cpu().reshape(8, 8).double()
            d[f"write_out_{nm}"] = round(r2[:, 3].mean().item() / max(r2[:, 6].mean().item(), 1), 1)
        out[f"{cin}->{cout}@{h}x{w}"] = d
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(d), flush=True)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino43_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
