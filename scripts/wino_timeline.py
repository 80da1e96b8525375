"""Cycle-level timeline of the Winograd forward kernel: the diag twin (variant 27 / 37) stamps s_memtime at the phase
boundaries of one mid-grid workgroup for chunks 8..23; this prints per-wave-group averages (cycles)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from tracknetv3_amd import ops
import diaglib


def main():
    dev = torch.device("cuda", 0)
    out = {}
    variants = [int(v) for v in (sys.argv[1:] or ["27"])]
    for cin, cout, h, w in ((512, 512, 36, 64), (256, 256, 72, 128), (64, 64, 288, 512)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        for v in variants:
            u = ops.pack_wino_weights(wt, variant=v)
            y = torch.zeros(10, cout, h, w, device=dev)
            for _ in range(3):
                diaglib.conv3x3_wino_forward(x, u, y, v)
            torch.cuda.synchronize()
            raw = y.view(-1)[: 8 * 8 * 2].cpu().numpy().view(np.uint64).reshape(8, 8).astype(np.int64)
            nch = int(raw[0, 6])
            per = raw[:, :6] / max(nch, 1)                     # cycles per chunk and phase, per wave
            d = {}
            names0 = ["loop_head", "dma_issue", "transform", "mfma", "own_dma_and_lds_done", "barrier_wait"]
            names1 = ["loop_head", "mfma", "transform", "-", "own_dma_and_lds_done", "barrier_wait"]
            for name, waves, names in (("grp0", slice(0, 4), names0), ("grp1", slice(4, 8), names1)):
                d[name] = {n: round(float(per[waves, i].mean()), 1) for i, n in enumerate(names) if n != "-"}
                d[name]["chunk_period"] = round(float(per[waves].sum(1).mean()), 1)
            d["chunks"] = nch
            out[f"{cout},{cin},{h},{w},v{v}"] = d
            print(f"{cout},{cin},{h}x{w} v{v}", json.dumps(d), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
