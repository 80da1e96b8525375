"""Times the tile configurations of conv_up2x on the three decoder-entry shapes (batch 10)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for name, c0, cout, hl, wl in (("up_block_1", 512, 256, 36, 64), ("up_block_2", 256, 128, 72, 128), ("up_block_3", 128, 64, 144, 256)):
        x = torch.rand(10, c0, hl, wl, device=dev)
        w = torch.rand(cout, c0 + c0 // 2, 3, 3, device=dev) - 0.5
        wq = ops.pack_up2x_weights(w, c0)
        gf = 2.0 * 4 * c0 * cout * (2 * hl) * (2 * wl) * 10 / 1e9
        row = {}
        for cfg in (0, 1, 2, 3):
            if cfg in (0, 2) and cout % 128:
                continue
            for _ in range(3):
                ops.conv_up2x(x, wq, cout, cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_up2x(x, wq, cout, cfg=cfg)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            row[cfg] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1)}
        out[name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
