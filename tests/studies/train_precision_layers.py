"""GPU study (round 6): WHICH LAYERS carry the training-mode heat-map error of the F(4x4) forward -- a finer Pareto than the per-level one
(train_precision_pareto.py: a whole level in F(2x2) form costs 0.25-1.3 ms; the cheapest 2x margin by levels costs +2.0 ms per step).

TrackNet(27, 8) at 288x512, N = 2, seed 31 (the chosen set also at seed 47), head gain 2.4 and 4.0, against the fp64 host oracle:
  1. every layer alone switched to F(2x2) (the others F(4x4)): how much that layer's F(4x4) rounding contributes;
  2. greedy: switch the layer with the best error reduction per step, until heat maps <= 3.5e-5 at gain 2.4 and <= 5e-5 at gain 4.0;
  3. milliseconds of a batch-10 training step for the prefixes of that sequence.
Layers in forward order: 0-1 down_block_1, 2-3 down_block_2, 4-6 down_block_3, 7-9 bottleneck, 10-12 up_block_1, 13-14 up_block_2, 15-16 up_block_3.
Prints one JSON object (gpurun_out/train_precision_layers.json).  Imports the oracle, hence lives under tests/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "studies"))
import torch  # noqa: E402

from oracle import nets  # noqa: E402
from tracknetv3_amd import tuning  # noqa: E402
from tracknetv3_amd.model import TrackNet  # noqa: E402
from train_precision_pareto import step_ms  # noqa: E402

GAINS = (2.4, 4.0)
BOUND = {2.4: 3.5e-5, 4.0: 5e-5}


def main():
    dev = torch.device("cuda:0")
    in_dim, out_dim, h, w, n = 27, 8, 288, 512, 2
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    seeds = (31, 47)
    oracle = {}
    for seed in seeds:
        for gain in GAINS:
            sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True, gain=gain)
            x = nets.synth_input((n, in_dim, h, w), seed + 1000)
            with torch.no_grad():
                sd64 = {k: (v.double().clone() if v.dtype != torch.int64 else v.clone()) for k, v in sd.items()}
                p64 = nets.tracknet_forward(sd64, x.double(), training=True)
            oracle[(seed, gain)] = (sd, x, p64)
            print("oracle", seed, gain, flush=True)

    def errs(f22_layers, seed=31):
        tuning.WINO43_TRAIN_F22_LAYERS = frozenset(f22_layers)
        out = {}
        for gain in GAINS:
            sd, x, p64 = oracle[(seed, gain)]
            m = TrackNet(in_dim, out_dim)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).train()
            with torch.no_grad():
                p = m(x.to(dev)).cpu().double()
            out[gain] = (p - p64).abs().max().item()
        return out

    def score(e):
        return max(e[g] / BOUND[g] for g in GAINS)

    res = {"bounds": {str(g): b for g, b in BOUND.items()}, "all_f43": errs(()), "all_f22": errs(range(17))}
    res["single_layer_to_f22"] = {str(i): errs((i,)) for i in range(17)}
    print(json.dumps(res), flush=True)
    chosen, seq = [], []
    cur = res["all_f43"]
    while score(cur) > 1.0 and len(chosen) < 17:
        best = min((i for i in range(17) if i not in chosen), key=lambda i: score(errs(chosen + [i])))
        chosen.append(best)
        cur = errs(chosen)
        seq.append({"added": best, "f22_layers": sorted(chosen), "heat_err": {str(g): cur[g] for g in GAINS}, "score": score(cur)})
        print(json.dumps(seq[-1]), flush=True)
    res["greedy"] = seq
    res["chosen"] = sorted(chosen)
    res["chosen_other_seed"] = errs(chosen, seed=47)
    res["all_f43_other_seed"] = errs((), seed=47)
    times = {}
    for k in sorted({0, len(chosen) // 2, len(chosen)}):
        tuning.WINO43_TRAIN_F22_LAYERS = frozenset(chosen[:k])
        times[str(k)] = round(min(step_ms(dev, (288, 144, 72, 36)), step_ms(dev, (288, 144, 72, 36))), 3)
    res["ms_per_step_batch10_by_prefix_length"] = times
    tuning.WINO43_TRAIN_F22_LAYERS = frozenset()
    print(json.dumps(res, indent=1))
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(res, open(os.path.join(od, "train_precision_layers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
