"""Robustness probe: does any kernel of the training step read memory it (or an earlier launch of the step) did not write?

torch.empty hands out whatever the caching allocator holds.  Fresh device pages are zero, so a kernel that reads an unwritten workspace,
statistics slot or halo usually gets zeros and passes every test -- until the cache holds the remains of something else (eight ranks on one
device, a long-running job).  This fills the allocator's cache with NaNs (blocks of every size class, freed again), runs the SAME training
step before and after, and compares loss, heat maps, BatchNorm buffers and all gradients bit for bit; with --locate it names the first
tensor that differs.  Shapes: the small one the data-parallel tests use (64x128: the fallback kernels of the deep levels) and 288x512.
  PARTS=custom CUSTOM_CMD="python scripts/train_poison_probe.py" bash scripts/gpu_session.sh"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tracknetv3_amd.utils import synth
from tracknetv3_amd.utils.general import get_model
from tracknetv3_amd.utils.metric import WBCELoss


def poison(dev, total_gb=6.0):
    """Leave NaN-filled blocks of every size class in the caching allocator."""
    keep, used = [], 0
    sizes = [1 << k for k in range(9, 31)]                      # 512 B .. 1 GiB
    for rep in range(3):
        for s in sizes:
            if used + s > total_gb * (1 << 30):
                continue
            t = torch.empty(s // 4, dtype=torch.float32, device=dev)
            t.fill_(float("nan"))
            keep.append(t)
            used += s
    torch.cuda.synchronize(dev)
    del keep
    return used


def step(net_fn, x, y, dev):
    net = net_fn().to(dev).train()
    p = net(x)
    loss = WBCELoss(p, y)
    loss.backward()
    torch.cuda.synchronize(dev)
    out = {"loss": loss.detach().cpu(), "heat": p.detach().cpu()}
    out.update({"grad/" + k: v.grad.detach().cpu() for k, v in net.named_parameters()})
    out.update({"buf/" + k: v.detach().cpu() for k, v in net.state_dict().items() if "running_" in k})
    return out


def main():
    dev = torch.device("cuda", 0)
    report = {}
    for tag, (seq, bg, n, h, w) in {"9to3@64x128": (3, "", 2, 64, 128), "27to8@288x512": (8, "concat", 2, 288, 512), "9to3@32x64": (3, "", 2, 32, 64)}.items():
        def net_fn():
            return synth.init_state_(get_model("TrackNet", seq, bg), 13, calibrated=True)
        in_dim = net_fn().in_dim
        g = torch.Generator().manual_seed(5)
        x = torch.rand((n, in_dim, h, w), generator=g).to(dev)
        y = synth.disc_heatmaps(n, seq, h, w, 77, device=dev)
        a = step(net_fn, x, y, dev)
        torch.cuda.empty_cache()
        used = poison(dev)
        b = step(net_fn, x, y, dev)
        c = step(net_fn, x, y, dev)                              # (and once more: the cache now holds the previous step's tensors)
        bad = [k for k in a if not (torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]))]
        nan = [k for k in b if torch.isnan(b[k]).any() or torch.isnan(c[k]).any()]
        report[tag] = {"poisoned_bytes": used, "tensors": len(a), "differ": bad[:12], "n_differ": len(bad), "nan": nan[:12],
                       "loss": [float(a["loss"]), float(b["loss"]), float(c["loss"])]}
        print(tag, json.dumps(report[tag]), flush=True)
        torch.cuda.empty_cache()
    # the eval forward the same way
    net = synth.init_state_(get_model("TrackNet", 8, "concat"), 13, calibrated=True).to(dev).eval()
    x = torch.rand((4, 27, 288, 512), device=dev)
    with torch.no_grad():
        y0 = net(x).cpu()
        torch.cuda.empty_cache()
        poison(dev)
        y1 = net(x).cpu()
    report["eval 27to8@288x512"] = {"equal": bool(torch.equal(y0, y1)), "nan": bool(torch.isnan(y1).any())}
    print("eval", json.dumps(report["eval 27to8@288x512"]), flush=True)
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(report, open(os.path.join(od, "train_poison_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
