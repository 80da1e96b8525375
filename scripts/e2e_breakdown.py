"""Stage breakdown of pipeline.predict_video on one GPU: wraps the stage functions with synchronising timers
(so the sum is an upper bound of the un-instrumented wall time) and prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tracknetv3_amd import pipeline, postprocess as pp, ops
from tracknetv3_amd.utils.general import get_model
from tracknetv3_amd.utils import synth


def main():
    dev = torch.device("cuda", 0)
    tn = get_model("TrackNet", 8, "concat")
    synth.init_state_(tn, 31, calibrated=True)
    tn = tn.to(dev).eval()
    net = get_model("InpaintNet").to(dev).eval()
    # bench.py's e2e leg: a synthetic 1080p uint8 stream resident in HBM (median + bicubic resize on the device are part of the flow)
    t_frames = int(os.environ.get("E2E_FRAMES", "256"))
    gen = torch.Generator(device=dev).manual_seed(99)
    bg = torch.randint(0, 96, (1, 1080, 1920, 3), dtype=torch.uint8, device=dev, generator=gen)
    frames = bg.repeat(t_frames, 1, 1, 1)
    for f in range(t_frames):
        cx, cy = 100 + 6 * (f % 280), 300 + (f * 7) % 500
        frames[f, cy - 8:cy + 9, cx - 8:cx + 9] = 255
    med = None
    acc = {}

    def timed(name, fn):
        def wrap(*a, **k):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(dev)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrap

    from tracknetv3_amd import preprocess
    orig = {(m, n): getattr(m, n) for m, n in ((preprocess, "median_background"), (preprocess, "resize_frames"), (pipeline, "_assemble"), (pp, "predict"),
                                                 (ops, "heatmap_peakfind"), (pp, "generate_inpaint_mask"), (pp, "inpaint_blend_threshold"))}
    preprocess.median_background = timed("median (1080p)", preprocess.median_background)
    preprocess.resize_frames = timed("resize (1080p -> 288x512)", preprocess.resize_frames)
    pipeline._assemble = timed("assemble", pipeline._assemble)
    pp.predict = timed("predict(host+peakfind)", pp.predict)
    ops.heatmap_peakfind = timed("  peakfind kernels", ops.heatmap_peakfind)
    pp.generate_inpaint_mask = timed("inpaint_mask", pp.generate_inpaint_mask)
    pp.inpaint_blend_threshold = timed("blend", pp.inpaint_blend_threshold)
    tn_fwd, net_fwd = tn.forward, net.forward
    tn.forward = timed("tracknet", tn_fwd)
    net.forward = timed("inpaintnet", net_fwd)
    out = {}
    for mode in ("nonoverlap", "weight"):
        pipeline.predict_video(frames[:40], tn, net, 8, 16, "concat", mode, 16)
        acc.clear()
        push = pp.EnsembleStream.push
        pp.EnsembleStream.push = timed("ensemble", push)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pipeline.predict_video(frames, tn, net, 8, 16, "concat", mode, 16)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        pp.EnsembleStream.push = push
        out[mode] = {"frames": t_frames, "instrumented_total_ms": round(dt * 1e3, 2), "instrumented_fps": round(t_frames / dt, 1),
                     "stages_ms": {k: round(v * 1e3, 2) for k, v in acc.items()},
                     "unaccounted_host_ms": round(dt * 1e3 - sum(v for k, v in acc.items() if not k.startswith("  ")) * 1e3, 2)}
    # the un-instrumented flow (what bench.py times): stages overlap, no synchronising timers
    for (m, n), f in orig.items():
        setattr(m, n, f)
    del tn.forward, net.forward                            # (the instance attributes that shadowed the class's forward)
    for mode in ("nonoverlap", "weight"):
        pipeline.predict_video(frames, tn, net, 8, 16, "concat", mode, 16)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            pipeline.predict_video(frames, tn, net, 8, 16, "concat", mode, 16)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        out[mode]["plain_total_ms"] = round(sorted(ts)[1] * 1e3, 2)
        out[mode]["plain_fps"] = round(t_frames / sorted(ts)[1], 1)
    print(json.dumps(out, indent=1))
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "e2e_breakdown.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
