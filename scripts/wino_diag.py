"""Where the Winograd kernel's time goes: the production kernel next to its timing twins (no DMA after the prologue /
+ no patch transform / + no barriers; results of the twins are wrong by design)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tracknetv3_amd import ops
import diaglib


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for cin, cout, h, w in ((512, 512, 36, 64), (256, 256, 72, 128), (64, 64, 288, 512)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        u = ops.pack_wino_weights((torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1)
        gf = 2.0 * 16 * cin * cout * (h // 2) * (w // 2) * 10 / 1e9
        row = {}
        for name, v in (("production", 0), ("no_dma", 11), ("no_dma_no_transform", 12), ("no_dma_no_transform_no_barrier", 13),
                        ("xisplit", 2), ("xisplit_dma_no_transform", 24), ("xisplit_dma_transform_no_writes", 25), ("xisplit_dma_transform_no_reads", 26), ("xisplit_no_dma", 21), ("xisplit_no_dma_no_transform", 22),
                        ("xisplit_no_dma_no_transform_no_barrier", 23)):
            y = torch.empty(10, cout, h, w, device=dev)
            for _ in range(3):
                diaglib.conv3x3_wino_forward(x, u, y, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                diaglib.conv3x3_wino_forward(x, u, y, v)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            row[name] = {"ms": round(ms, 4), "executed_tflops": round(gf / ms, 1)}
        out[f"{cout},{cin},10,{h},{w}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
