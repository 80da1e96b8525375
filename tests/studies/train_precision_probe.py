"""GPU study (round 5): WHERE the training-mode heat-map error of the F(4x4) forward comes from.

The full-size N = 2 training forward of TrackNet(27, 8) against the fp64 host oracle with the F(4x4) statistics-epilogue kernel
enabled only at ONE resolution level (288 / 144 / 72 / 36 rows), at all of them (the default), at none (F(2x2) everywhere), and with
the 25-of-36 form of the decoder entries' upsampled halves on top of the default.  Prints one JSON object; run by
scripts/gpu_session.sh PARTS=custom.  Imports the oracle, hence lives under tests/.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import nets  # noqa: E402
from tracknetv3_amd import tuning  # noqa: E402
from tracknetv3_amd.model import TrackNet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    in_dim, out_dim, h, w, seed, n = 27, 8, 288, 512, 31, 2
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True)
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    with torch.no_grad():
        sd64 = {k: (v.double().clone() if v.dtype != torch.int64 else v.clone()) for k, v in sd.items()}
        p64 = nets.tracknet_forward(sd64, x.double(), training=True)
        p32 = nets.tracknet_forward({k: v.clone() for k, v in sd.items()}, x, training=True).double()
    out = {"torch_fp32": (p32 - p64).abs().max().item()}
    real = tuning.use_wino43_train

    def run(tag, levels, up_variant=0):
        tuning.use_wino43_train = (lambda cin, cout, hh, ww, layer=None: hh in levels and real(cin, cout, hh, ww))
        tuning.UP2X_WINO_VARIANT_TRAIN = up_variant
        m = TrackNet(in_dim, out_dim)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        with torch.no_grad():
            p = m(x.to(dev)).cpu().double()
        out[tag] = (p - p64).abs().max().item()

    run("f22_everywhere", ())
    for lv in (288, 144, 72, 36):
        run(f"f43_only_at_{lv}", (lv,))
    run("f43_everywhere(default)", (288, 144, 72, 36))
    run("f43_everywhere+up2x_25of36", (288, 144, 72, 36), 2)
    run("f22_everywhere+up2x_25of36", (), 2)
    tuning.use_wino43_train = real
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
