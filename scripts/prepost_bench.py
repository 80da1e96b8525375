"""The three byte / integer stages either side of TrackNet in predict_video (1080p source, BASELINE configs[4]) timed alone on one GPU: temporal
median over T frames, Pillow-exact bicubic resize to 288x512 (both passes), heat-map peak-find -- ms per call and the effective HBM rate of
the bytes each MUST move (median: T x P read once + P written; resize: source read + fp32 CHW written; peak-find: the maps read).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  argv: T (default 256)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import ops, preprocess


def timeit(fn, reps=5):
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
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(99)
    bg = torch.randint(0, 96, (1, 1080, 1920, 3), dtype=torch.uint8, device=dev, generator=gen)
    frames = bg.repeat(t, 1, 1, 1)
    frames += torch.randint(0, 8, frames.shape, dtype=torch.uint8, device=dev, generator=gen)      # (per-frame noise: the median has something to select)
    for f in range(t):
        cx, cy = 100 + 6 * (f % 280), 300 + (f * 7) % 500
        frames[f, cy - 8:cy + 9, cx - 8:cx + 9] = 255
    out = {"frames": t}
    dst = torch.empty_like(frames)
    ms = timeit(lambda: dst.copy_(frames))
    out["plain_copy_of_the_stack"] = {"ms": round(ms, 3), "GB_read_plus_written": round(2 * frames.numel() / 1e9, 3), "TBps": round(2 * frames.numel() / ms / 1e9, 2)}
    del dst
    ms = timeit(lambda: preprocess.median_background(frames))
    out["median_1080p"] = {"ms": round(ms, 3), "must_move_GB": round((t + 1) * 1080 * 1920 * 3 / 1e9, 3), "TBps": round((t + 1) * 1080 * 1920 * 3 / ms / 1e9, 2)}
    ms = timeit(lambda: preprocess.resize_frames(frames))
    gb = t * (1080 * 1920 * 3 + 288 * 512 * 3 * 4) / 1e9
    out["resize_1080p_to_288x512_f32"] = {"ms": round(ms, 3), "must_move_GB": round(gb, 3), "TBps": round(gb / ms, 2)}
    heat = torch.zeros((t // 2, 288, 512), device=dev)
    for f in range(t // 2):                                  # a ball-sized blob per map + a few specks, like the network's output
        cx, cy = 20 + (f * 3) % 470, 20 + (f * 5) % 250
        heat[f, cy - 3:cy + 4, cx - 3:cx + 4] = 0.9
        heat[f, (cy * 7) % 280, (cx * 3) % 500] = 0.7
    ms = timeit(lambda: ops.heatmap_peakfind(heat))
    out["peakfind_%d_maps" % (t // 2)] = {"ms": round(ms, 3), "must_move_GB": round(heat.numel() * 4 / 1e9, 3), "TBps": round(heat.numel() * 4 / ms / 1e9, 2)}
    print(json.dumps(out, indent=1))
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "prepost_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
