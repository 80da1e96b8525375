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
    t_frames = 264
    frames = torch.rand(t_frames, 3, 288, 512, device=dev) * 0.2
    for f in range(t_frames):
        cx, cy = 20 + f, 60 + (f * 3) % 150
        frames[f, :, cy - 2:cy + 3, cx - 2:cx + 3] = 1.0
    med = frames.median(dim=0).values
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
        pipeline.predict_video(frames[:40], tn, net, 8, 16, "concat", mode, 16, (1920, 1080), median=med)
        acc.clear()
        push = pp.EnsembleStream.push
        pp.EnsembleStream.push = timed("ensemble", push)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pipeline.predict_video(frames, tn, net, 8, 16, "concat", mode, 16, (1920, 1080), median=med)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        pp.EnsembleStream.push = push
        out[mode] = {"total_ms": round(dt * 1e3, 2), "fps": round(t_frames / dt, 1),
                     "stages_ms": {k: round(v * 1e3, 2) for k, v in acc.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
