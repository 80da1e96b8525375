"""Soak run on the GPU box: 80 batch-10 training steps (memory must not grow, the loss must fall) and 8 predict_video calls
(reproducible across calls).  python scripts/soak.py"""
import sys, torch, time
sys.path.insert(0, ".")


def main():
    from tracknetv3_amd.utils.general import get_model
    from tracknetv3_amd.parallel import TrackNetTrainer
    from tracknetv3_amd.pipeline import predict_video
    dev = torch.device("cuda", 0)
    net = get_model("TrackNet", 8, "concat").to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    tr = TrackNetTrainer(net, opt, alpha=0.5)
    x = torch.rand(10, 27, 288, 512, device=dev); y = (torch.rand(10, 8, 288, 512, device=dev) > 0.999).float()
    losses = []
    for i in range(80):
        losses.append(tr.step(x, y))
        if i in (4, 29, 54, 79):
            torch.cuda.synchronize(); print("train step", i, "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), "loss", float(losses[-1]))
    assert all(torch.isfinite(l) for l in losses) and float(losses[-1]) < float(losses[0])
    net.eval(); inp = get_model("InpaintNet").to(dev).eval()
    frames = torch.rand(200, 3, 288, 512, device=dev) * 0.2
    ref = None
    for i in range(8):
        pd = predict_video(frames, net, inp, 8, 16, "concat", "weight" if i % 2 else "nonoverlap", 16)
        key = (i % 2, tuple(pd["X"]), tuple(pd["Y"]))
        if i < 2: ref = {**(ref or {}), i % 2: key}
        else: assert ref[i % 2] == key, "predict_video not reproducible across calls"
        if i in (1, 7):
            torch.cuda.synchronize(); print("predict call", i, "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2))
    print("soak ok")


if __name__ == "__main__":
    main()
