"""Peak-find (threshold + connected components + largest box) timing on representative and adversarial heat maps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops


def main():
    dev = torch.device("cuda", 0)
    frames, h, w = 128, 288, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    maps = {}
    sparse = torch.zeros(frames, h, w)
    for f in range(frames):
        cx, cy = 20 + 3 * f, 40 + (f * 7) % 200
        sparse[f, cy - 3:cy + 4, cx - 3:cx + 4] = 0.9
    maps["one_blob_7x7"] = sparse
    maps["noise_p50"] = torch.rand(frames, h, w, generator=g)
    maps["noise_p05"] = torch.rand(frames, h, w, generator=g) * 0.526
    maps["all_foreground"] = torch.ones(frames, h, w)
    maps["empty"] = torch.zeros(frames, h, w)
    out = {}
    for name, m in maps.items():
        m = m.to(dev).contiguous()
        ops.heatmap_peakfind(m)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.heatmap_peakfind(m)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / 5
        out[name] = {"ms_per_128_frames": round(ms, 3), "frames_per_s": round(frames / ms * 1e3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
