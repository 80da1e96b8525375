"""Synthetic workloads for benchmarks and soak runs (no datasets or checkpoints are reachable): seeded weights of the
reference's architecture and labels of the reference's format.  Product-side on purpose -- `oracle/` is test
infrastructure and must not be needed to run the hot path.

Values follow the reference where it defines them: conv / linear weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the PyTorch
default bound the reference relies on, model.py:8,54), labels are binary discs of radius 2.5 (dataset.py:401-410)."""
import math

import torch


def init_state_(module, seed, calibrated=True):
    """Fill every parameter / buffer of a TrackNet or InpaintNet module in place, deterministically from `seed` (a CPU
    generator: the same values on every device).  calibrated: non-trivial BatchNorm gamma / beta / running statistics and a
    weight gain of 2.4, which keeps the activations O(1) through the 17 ReLU layers (random weights at the default bound
    shrink them layer by layer); otherwise fresh-module BatchNorm values."""
    g = torch.Generator().manual_seed(int(seed))

    def uniform(shape, lo, hi):
        return torch.rand(tuple(shape), generator=g, dtype=torch.float32) * (hi - lo) + lo

    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if t.dtype == torch.int64:
                t.zero_()
            elif name.endswith("bn.weight"):
                t.copy_(uniform(t.shape, 0.5, 1.5) if calibrated else torch.ones(t.shape))
            elif name.endswith("bn.bias"):
                t.copy_(uniform(t.shape, -0.3, 0.3) if calibrated else torch.zeros(t.shape))
            elif name.endswith("running_mean"):
                t.copy_(uniform(t.shape, -0.2, 0.2) if calibrated else torch.zeros(t.shape))
            elif name.endswith("running_var"):
                t.copy_(uniform(t.shape, 0.5, 2.0) if calibrated else torch.ones(t.shape))
            elif name.endswith(".weight"):
                b = 1.0 / math.sqrt(int(math.prod(t.shape[1:])))
                gain = 2.4 if calibrated else 1.0
                t.copy_(uniform(t.shape, -b * gain, b * gain))
            elif name.endswith(".bias"):
                w = sd[name[:-4] + "weight"]
                b = 1.0 / math.sqrt(int(math.prod(w.shape[1:])))
                t.copy_(uniform(t.shape, -b, b))
            else:
                raise KeyError(name)
    return module


def disc_heatmaps(n, seq_len, h, w, seed, device="cpu", sigma=2.5):
    """(n, seq_len, h, w) fp32 targets: pixel (i, j) is 1 iff (i - cy)^2 + (j - cx)^2 <= sigma^2 around a random centre;
    every fifth map is empty (an invisible shuttlecock)."""
    g = torch.Generator().manual_seed(int(seed))
    c = torch.rand((n, seq_len, 2), generator=g)
    cx = (c[..., 0] * w).floor().view(n, seq_len, 1, 1)
    cy = (c[..., 1] * h).floor().view(n, seq_len, 1, 1)
    ii = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1)
    jj = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w)
    y = (((ii - cy) ** 2 + (jj - cx) ** 2) <= sigma ** 2).float()
    empty = (torch.arange(n * seq_len).view(n, seq_len) % 5 == 4) | ((cx.view(n, seq_len) == 0) & (cy.view(n, seq_len) == 0))
    y[empty] = 0.0
    return y.to(device)
