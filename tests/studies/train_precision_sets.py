"""GPU study (round 6): verification of candidate training-forward configurations over seeds and gains.

For each configuration -- a set of layers whose training forward runs in F(2x2) form (the others F(4x4)), with the decoder entries' upsampled
halves in the 9-GEMM F(2x2) form or the 25-of-36 F(4x4) form -- the training-mode heat-map error of TrackNet(27, 8) at 288x512, N = 2, against
the fp64 host oracle for seeds 31 / 47 / 59 x head gain 2.4 / 4.0 / 6.0, and the milliseconds of a batch-10 training step.
CONFIGS (env): ';'-separated "layers|up" entries, e.g. "|0;0,1|0;0,1|2".  Prints one JSON object (gpurun_out/train_precision_sets.json)."""
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

SEEDS, GAINS = (31, 47, 59), (2.4, 4.0, 6.0)


def main():
    dev = torch.device("cuda:0")
    in_dim, out_dim, h, w, n = 27, 8, 288, 512, 2
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    cfgs = []
    for ent in os.environ.get("CONFIGS", "|0;0|0;0,1|0;0,1|2;|2").split(";"):
        layers, up = ent.split("|")
        cfgs.append((tuple(int(v) for v in layers.split(",") if v.strip()), int(up)))
    oracle = {}
    for seed in SEEDS:
        for gain in GAINS:
            sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True, gain=gain)
            x = nets.synth_input((n, in_dim, h, w), seed + 1000)
            with torch.no_grad():
                sd64 = {k: (v.double().clone() if v.dtype != torch.int64 else v.clone()) for k, v in sd.items()}
                p64 = nets.tracknet_forward(sd64, x.double(), training=True)
                p32 = nets.tracknet_forward({k: v.clone() for k, v in sd.items()}, x, training=True).double()
            oracle[(seed, gain)] = (sd, x, p64, (p32 - p64).abs().max().item())
            print("oracle", seed, gain, flush=True)
    out = {"seeds": list(SEEDS), "gains": list(GAINS), "torch_fp32": {f"seed{s}_gain{g}": oracle[(s, g)][3] for s, g in oracle}, "configs": {}}
    up_default = tuning.UP2X_WINO_VARIANT_TRAIN
    for layers, up in cfgs:
        tuning.WINO43_TRAIN_F22_LAYERS = frozenset(layers)
        tuning.UP2X_WINO_VARIANT_TRAIN = up
        row = {}
        for (seed, gain), (sd, x, p64, _) in oracle.items():
            m = TrackNet(in_dim, out_dim)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).train()
            with torch.no_grad():
                p = m(x.to(dev)).cpu().double()
            row[f"seed{seed}_gain{gain}"] = (p - p64).abs().max().item()
        row["worst_by_gain"] = {str(g): max(row[f"seed{s}_gain{g}"] for s in SEEDS) for g in GAINS}
        row["ms_per_step_batch10"] = round(min(step_ms(dev, (288, 144, 72, 36)), step_ms(dev, (288, 144, 72, 36))), 3)
        tag = f"f22_layers[{','.join(str(v) for v in layers)}]_up2x_form{up}"
        out["configs"][tag] = row
        print(tag, json.dumps(row["worst_by_gain"]), row["ms_per_step_batch10"], flush=True)
    tuning.WINO43_TRAIN_F22_LAYERS = frozenset()
    tuning.UP2X_WINO_VARIANT_TRAIN = up_default
    print(json.dumps(out, indent=1))
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, "train_precision_sets.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
