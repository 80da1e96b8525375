"""Cycle-level timeline of the 128-channel Winograd kernel (variant 6): its diag twin 107 stamps s_memtime at the phase boundaries of
every chunk of one mid-grid workgroup; this prints per-wave-group averages (cycles per chunk / per tile)."""
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
    twin = int(sys.argv[1]) if len(sys.argv) > 1 else 107        # 107: production order, 105: group 0 (older waves) runs its MFMAs first
    for cin, cout, h, w in ((512, 512, 36, 64), (256, 256, 72, 128), (128, 128, 144, 256), (64, 128, 144, 256)):
        x = torch.relu(torch.randn(10, cin, h, w, device=dev))
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        u = ops.pack_wino_weights(wt, variant=6)
        y = torch.zeros(11, cout, h, w, device=dev)
        for _ in range(3):
            diaglib.conv3x3_wino_forward(x, u, y, twin)
        torch.cuda.synchronize()
        raw = y[10].reshape(-1)[: 8 * 10 * 2].cpu().numpy().view(np.uint64).reshape(8, 10).astype(np.int64)
        nch, ntile = int(raw[0, 8]), int(raw[0, 9])
        d = {"chunks": nch, "tiles": ntile}
        names0 = ["head", "patch_reads+dma", "transform", "mfma", "own_dma_and_lds_done", "barrier_wait"]
        names1 = ["head", "mfma", "patch_reads+dma", "transform", "own_dma_and_lds_done", "barrier_wait"]
        if twin in (105, 113):
            names0, names1 = names1, names0
        for name, waves, names in (("grp0", slice(0, 4), names0), ("grp1", slice(4, 8), names1)):
            per = raw[waves, :6] / max(nch, 1)
            d[name] = {n: round(float(per[:, i].mean()), 1) for i, n in enumerate(names)}
            d[name]["chunk_period"] = round(float(per.sum(1).mean()), 1)
            d[name]["writeout_per_tile"] = round(float(raw[waves, 6].mean()) / max(ntile, 1), 1)
            d[name]["advance_per_tile"] = round(float(raw[waves, 7].mean()) / max(ntile, 1), 1)
        out[f"{cin}->{cout}@{h}x{w}"] = d
        print(f"{cin}->{cout}@{h}x{w}", json.dumps(d), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wino6_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
