"""CPU checks of bench.py's bookkeeping (no GPU): the layer table reproduces SURVEY 8d's algorithmic FLOP count, the executed
count is consistent with which layers the reformulated kernels take, and the CLI keeps the driver's contract flags."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_layer_table_matches_the_algorithmic_flop_count():
    b = _bench()
    layers = b.conv_layer_table(27, 288, 512)
    assert len(layers) == 17 and sum(1 for l in layers if l[6]) == 3
    total = sum(b.conv_flops(c0, c1, co, h, w) for (_, c0, c1, co, h, w, _) in layers)
    assert abs(total / 1e9 - 227.455) < 0.01                      # 227.606 GFLOP/sample incl. the 0.151 GFLOP head
    # the three decoder-entry layers hold 2/3 of their inputs in the upsampled tensor
    for (_, c0, c1, co, h, w, up) in layers:
        if up:
            assert c0 == 2 * c1 and b.conv_flops(c0, 0, co, h, w) == 2 * b.conv_flops(c1, 0, co, h, w)


def test_training_flop_constants():
    b = _bench()
    assert abs(b.TRAIN_FLOPS_PER_SAMPLE / 1e9 - 678.2) < 0.1
    up = 3 * 21.743                                               # upsampled halves, each pass
    plain = 9 * 10.872 + 4 * 5.436                                # the 13 plain layers with >= 64 channels
    fwd = 4.586 * 16 / 36 + plain * 16 / 36 + up * 9 / 36 + 3 * 10.872 * 16 / 36   # upsampled halves: 9 of the 16 Winograd GEMMs
    dgrad = plain * 16 / 36 + up * 9 / 36 + 3 * 10.872 * 16 / 36
    wgrad = 4.586 + plain * 16 / 36 + up * 9 / 36 + 3 * 10.872 * 16 / 36     # only the first layer keeps the direct kernel
    assert abs((fwd + dgrad + wgrad) - b.TRAIN_FLOPS_EXECUTED_PER_SAMPLE / 1e9) < 1.5
    assert abs((fwd + dgrad + wgrad) + 3 * up * (4 / 9 - 9 / 36) - b.TRAIN_FLOPS_EXECUTED_PER_SAMPLE_CLASS_FILTERS / 1e9) < 1.5


def test_cli_contract_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
                '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src
