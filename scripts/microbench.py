#!/usr/bin/env python
"""Secondary measurements on one MI355X (BASELINE configs[3] and the post-process of configs[4]):
InpaintNet forward latency/throughput, batched peak-find, temporal ensemble.  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, reps, dev):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def main():
    from tracknetv3_amd import ops, postprocess as pp
    from tracknetv3_amd.utils.general import get_model
    dev = torch.device("cuda:0")
    out = {}
    # sustained fp32 MFMA rate of this chip under load (register-only probe): the practical ceiling of the conv kernels
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import diaglib
    for blocks, label in ((256 * 1, "1wg_per_cu"), (256 * 3, "3wg_per_cu"), (256 * 8, "8wg_per_cu")):
        buf = torch.empty(blocks * 256, device=dev)
        iters = 4000
        ms = timeit(lambda: diaglib.mfma_f32_probe(buf, blocks, iters), 5, dev)
        flops = blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 2)
        out[f"mfma_f32_probe_{label}"] = {"ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1)}
    net = get_model("InpaintNet").to(dev).eval()
    from tracknetv3_amd import inpaint_ops
    for n in (1, 32, 256, 512, 2048, 8192, 65536):
        x, m = torch.rand(n, 16, 2, device=dev), (torch.rand(n, 16, 1, device=dev) < 0.3).float()
        for tag, flag in (("fused", "1"), ("layers", "0")):
            inpaint_ops.FUSED = flag
            ms = timeit(lambda: net(x, m), 5 if n > 4096 else 20, dev)
            out[f"inpaintnet_fwd_n{n}_{tag}"] = {"ms": round(ms, 4), "seq_per_s": round(n / ms * 1e3, 1), "tflops": round(16.63e6 * n / ms / 1e9, 2)}
    inpaint_ops.FUSED = "auto"
    x, m = torch.rand(32, 16, 2, device=dev), (torch.rand(32, 16, 1, device=dev) < 0.3).float()
    hm = torch.zeros(80, 288, 512, device=dev)
    hm[:, 100:106, 200:207] = 0.9
    hm[::3, 20:23, 400:404] = 0.8
    out["peakfind_80x288x512_sparse"] = {"ms": round(timeit(lambda: ops.heatmap_peakfind(hm, 0.5), 20, dev), 4)}
    hd = (torch.rand(80, 288, 512, device=dev) < 0.5).float()
    out["peakfind_80x288x512_dense_noise"] = {"ms": round(timeit(lambda: ops.heatmap_peakfind(hd, 0.5), 5, dev), 4)}
    win = torch.rand(17, 8, 288, 512, device=dev)
    w = pp.get_ensemble_weight(8, "weight").to(dev)
    ms = timeit(lambda: ops.ensemble_frames(win, 0, w, 7, 10, 1000), 20, dev)
    out["ensemble_10frames_L8"] = {"ms": round(ms, 4), "GBps": round((10 * 9) * 288 * 512 * 4 / ms / 1e6, 1)}
    x = torch.rand(10, 27, 288, 512, device=dev)
    lam, perm = torch.rand(10, device=dev), torch.randperm(10, device=dev).int()
    ms = timeit(lambda: ops.mixup(x, lam, perm), 20, dev)
    out["mixup_10x27x288x512"] = {"ms": round(ms, 4), "GBps": round(3 * x.numel() * 4 / ms / 1e6, 1)}
    # ---- BASELINE configs[4]: end-to-end predict.py path on a synthetic stream (frames already resized to 288x512;
    #      video decode / bicubic resize / median are outside the path)
    from tracknetv3_amd.utils import synth
    from tracknetv3_amd.pipeline import predict_video
    tn = get_model("TrackNet", 8, "concat")
    synth.init_state_(tn, 31, calibrated=True)
    tn = tn.to(dev).eval()
    t_frames = 264
    frames = torch.rand(t_frames, 3, 288, 512, device=dev) * 0.2
    for f in range(t_frames):
        cx, cy = 20 + f, 60 + (f * 3) % 150
        frames[f, :, cy - 2:cy + 3, cx - 2:cx + 3] = 1.0
    med = frames.median(dim=0).values
    for mode in ("nonoverlap", "weight"):
        predict_video(frames, tn, net, 8, 16, "concat", mode, 16, (1920, 1080), median=med)           # warm-up (same batch shapes: allocator pools of both streams)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pd = predict_video(frames, tn, net, 8, 16, "concat", mode, 16, (1920, 1080), median=med)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        assert len(pd["Frame"]) == t_frames
        out[f"e2e_predict_video_{mode}"] = {"frames": t_frames, "s": round(dt, 4), "fps": round(t_frames / dt, 1),
                                            "note": "TrackNet(8,concat)+InpaintNet(16), batch 16, frames resident at 288x512"}
    # ---- device-side preprocessing and the end-to-end path from 1080p uint8 frames
    from tracknetv3_amd import preprocess as pre
    src = torch.randint(0, 256, (64, 1080, 1920, 3), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: pre.resize_frames(src), 5, dev)
    out["resize_1080p_to_288x512_64frames"] = {"ms": round(ms, 3), "frames_per_s": round(64 / ms * 1e3, 1),
                                                "GBps_read": round(src.numel() / ms / 1e6, 1)}
    ms = timeit(lambda: pre.median_background(src), 3, dev)
    out["median_1080p_T64"] = {"ms": round(ms, 3), "GBps_read": round(src.numel() / ms / 1e6, 1)}
    big = src.repeat(4, 1, 1, 1)[:256]
    for f in range(256):
        cx, cy = 100 + 6 * f, 300 + (f * 7) % 500
        big[f, cy - 8:cy + 9, cx - 8:cx + 9] = 255
    for mode in ("nonoverlap", "weight"):
        predict_video(big, tn, net, 8, 16, "concat", mode, 16)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pd = predict_video(big, tn, net, 8, 16, "concat", mode, 16)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out[f"e2e_from_1080p_u8_{mode}"] = {"frames": 256, "s": round(dt, 4), "fps": round(256 / dt, 1),
                                             "note": "median + bicubic resize + TrackNet + ensemble + peak-find + InpaintNet; frames resident as uint8 1080p"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
