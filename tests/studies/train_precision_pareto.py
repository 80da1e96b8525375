"""GPU study (round 6): the error-vs-milliseconds Pareto of WHERE the training forward runs in F(4x4) form.

For every subset of the four resolution levels (288 / 144 / 72 / 36 rows) at which the F(4x4) statistics-epilogue kernel may replace the
F(2x2) one: the training-mode heat-map error of TrackNet(27, 8) at 288x512, N = 2, against the fp64 host oracle at head gain 2.4 / 4 / 6
(seed 31; the chosen sets also at seed 47), and the milliseconds of a batch-10 training step with that subset.  The product default
(tuning.WINO43_TRAIN_LEVELS) is the cheapest subset with <= 3.5e-5 at gain 2.4 and <= 5e-5 at gain 4 (2x under north_star's 1e-4 at the
trained-like logit range).  Prints one JSON object; run through scripts/gpu_session.sh PARTS=custom.  Imports the oracle, hence lives under tests/.
"""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import nets  # noqa: E402
from tracknetv3_amd import tuning  # noqa: E402
from tracknetv3_amd.model import TrackNet  # noqa: E402

LEVELS = (288, 144, 72, 36)
GAINS = (2.4, 4.0, 6.0)


def step_ms(dev, levels, steps=8, warmup=3):
    from tracknetv3_amd.optim import FusedAdam
    from tracknetv3_amd.parallel import TrackNetTrainer
    from tracknetv3_amd.utils import synth
    tuning.WINO43_TRAIN_LEVELS = frozenset(levels)
    net = synth.init_state_(TrackNet(27, 8), 13, calibrated=True).to(dev)
    tr = TrackNetTrainer(net, FusedAdam(net.parameters(), lr=1e-3), alpha=0.5)
    g = torch.Generator().manual_seed(5)
    x = torch.rand((10, 27, 288, 512), generator=g).to(dev)
    y = synth.disc_heatmaps(10, 8, 288, 512, 77, device=dev)
    for _ in range(warmup):
        tr.step(x, y)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        tr.step(x, y)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps


def main():
    dev = torch.device("cuda:0")
    in_dim, out_dim, h, w, n = 27, 8, 288, 512, 2
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    subsets = [tuple(s) for k in range(5) for s in itertools.combinations(LEVELS, k)]
    seeds = [int(v) for v in os.environ.get("PARETO_SEEDS", "31").split()]
    default_levels = tuning.WINO43_TRAIN_LEVELS
    out = {"levels": list(LEVELS), "gains": list(GAINS), "seeds": seeds, "rows": {}}
    oracle = {}
    for seed in seeds:
        for gain in GAINS:
            sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True, gain=gain)
            x = nets.synth_input((n, in_dim, h, w), seed + 1000)
            with torch.no_grad():
                sd64 = {k: (v.double().clone() if v.dtype != torch.int64 else v.clone()) for k, v in sd.items()}
                p64 = nets.tracknet_forward(sd64, x.double(), training=True)
                p32 = nets.tracknet_forward({k: v.clone() for k, v in sd.items()}, x, training=True).double()
            oracle[(seed, gain)] = (sd, x, p64, (p32 - p64).abs().max().item(), [p64.min().item(), p64.max().item()])
            print(f"oracle seed {seed} gain {gain}: torch-fp32 {oracle[(seed, gain)][3]:.3e}, heat-map range {oracle[(seed, gain)][4]}", flush=True)
    out["torch_fp32"] = {f"seed{seed}_gain{gain}": oracle[(seed, gain)][3] for seed, gain in oracle}
    out["heat_range"] = {f"seed{seed}_gain{gain}": oracle[(seed, gain)][4] for seed, gain in oracle}
    for levels in subsets:
        tag = "+".join(str(v) for v in levels) or "none(F(2x2) everywhere)"
        row = {}
        tuning.WINO43_TRAIN_LEVELS = frozenset(levels)
        for (seed, gain), (sd, x, p64, _, _) in oracle.items():
            m = TrackNet(in_dim, out_dim)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).train()
            with torch.no_grad():
                p = m(x.to(dev)).cpu().double()
            row[f"heat_err_seed{seed}_gain{gain}"] = (p - p64).abs().max().item()
        row["ms_per_step_batch10"] = round(min(step_ms(dev, levels), step_ms(dev, levels)), 3)
        out["rows"][tag] = row
        print(tag, json.dumps(row), flush=True)
    tuning.WINO43_TRAIN_LEVELS = default_levels
    ok = {t: r for t, r in out["rows"].items()
          if max(r[f"heat_err_seed{s_}_gain2.4"] for s_ in seeds) <= 3.5e-5 and max(r[f"heat_err_seed{s_}_gain4.0"] for s_ in seeds) <= 5e-5}
    out["cheapest_with_2x_margin"] = min(ok, key=lambda t: ok[t]["ms_per_step_batch10"]) if ok else None
    print(json.dumps(out, indent=1))
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "train_precision_pareto.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
