"""CPU restatement of the TrackNetV3 network arithmetic (TrackNet, InpaintNet, WBCE).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never imported by the
product package.

Every function is a *functional* restatement over a plain ``state_dict``
(name -> tensor) written with elementary torch CPU ops, so that it can run in
fp32 or fp64 and be differentiated by autograd.  BatchNorm is written out
explicitly (not ``nn.BatchNorm2d``).  Reference lines followed (paths relative
to the upstream repository):

* Conv2DBlock  conv3x3(no bias, zero pad 1) -> BN(eps 1e-5, momentum .1) -> ReLU   model.py:4-16
* Double2DConv / Triple2DConv                                                        model.py:18-42
* TrackNet.__init__ channel plan, predictor 1x1 + bias                               model.py:45-55
* TrackNet.forward  pool / nearest-upsample / cat([up, skip]) / sigmoid              model.py:57-73
* Conv1DBlock  conv1d k3 pad1 + bias -> LeakyReLU(0.01)                              model.py:76-87
* InpaintNet.forward  cat([x, m], 2) -> permute -> ... -> sigmoid -> permute         model.py:113-129
* WBCELoss                                                                           utils/metric.py:3-20

Parity pinning: ``tests/golden/make_golden.py`` (run where the reference is
importable) checks these functions against the imported reference modules on
the same weights/inputs and commits the resulting vectors under tests/golden/.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import prng

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (block name, number of Conv2DBlocks, in channels or None (=network input), out channels)
_TRACKNET_PLAN = (
    ("down_block_1", 2, None, 64),
    ("down_block_2", 2, 64, 128),
    ("down_block_3", 3, 128, 256),
    ("bottleneck", 3, 256, 512),
    ("up_block_1", 3, 768, 256),
    ("up_block_2", 2, 384, 128),
    ("up_block_3", 2, 192, 64),
)


def tracknet_dims(seq_len, bg_mode):
    """(in_dim, out_dim) exactly as get_model does -- utils/general.py:66-74."""
    if bg_mode == "subtract":
        return seq_len, seq_len
    if bg_mode == "subtract_concat":
        return seq_len * 4, seq_len
    if bg_mode == "concat":
        return (seq_len + 1) * 3, seq_len
    return seq_len * 3, seq_len


def tracknet_conv_layers(in_dim):
    """[(prefix, cin, cout)] for the 17 Conv2DBlocks in forward order."""
    out = []
    for blk, n, cin, cout in _TRACKNET_PLAN:
        c = in_dim if cin is None else cin
        for k in range(1, n + 1):
            out.append((f"{blk}.conv_{k}", c, cout))
            c = cout
    return out


def tracknet_state_shapes(in_dim, out_dim):
    """OrderedDict name -> (shape, dtype) in nn.Module.state_dict() order."""
    d = OrderedDict()
    for p, cin, cout in tracknet_conv_layers(in_dim):
        d[f"{p}.conv.weight"] = ((cout, cin, 3, 3), torch.float32)
        d[f"{p}.bn.weight"] = ((cout,), torch.float32)
        d[f"{p}.bn.bias"] = ((cout,), torch.float32)
        d[f"{p}.bn.running_mean"] = ((cout,), torch.float32)
        d[f"{p}.bn.running_var"] = ((cout,), torch.float32)
        d[f"{p}.bn.num_batches_tracked"] = ((), torch.int64)
    d["predictor.weight"] = ((out_dim, 64, 1, 1), torch.float32)
    d["predictor.bias"] = ((out_dim,), torch.float32)
    return d


_INPAINT_LAYERS = (
    ("down_1.conv", 3, 32), ("down_2.conv", 32, 64), ("down_3.conv", 64, 128),
    ("buttleneck.conv_1.conv", 128, 256), ("buttleneck.conv_2.conv", 256, 256),
    ("up_1.conv", 384, 128), ("up_2.conv", 192, 64), ("up_3.conv", 96, 32),
    ("predictor", 32, 2),
)


def inpaintnet_state_shapes():
    d = OrderedDict()
    for p, cin, cout in _INPAINT_LAYERS:
        d[f"{p}.weight"] = ((cout, cin, 3), torch.float32)
        d[f"{p}.bias"] = ((cout,), torch.float32)
    return d


def synth_state(shapes, seed, calibrated=False, gain=None, var_range=None):
    """Deterministic synthetic state_dict from the portable PRNG.

    Conv/linear weights ~ U(-b, b) with b = 1/sqrt(fan_in) (the PyTorch default
    bound).  ``calibrated=False``: BN gamma=1, beta=0, rm=0, rv=1
    (fresh-module values).  ``calibrated=True``: non-trivial gamma/beta/rm/rv so
    that the BN arithmetic is actually exercised.  ``gain`` / ``var_range``
    (calibrated only; defaults 2.4 and (0.5, 2.0), the values every golden fixture
    was made with) are what the precision sweep varies: the conv weights' gain and
    the range the BN running variances are drawn from.
    """
    gain_cal = 2.4 if gain is None else float(gain)
    v_lo, v_hi = (0.5, 2.0) if var_range is None else (float(var_range[0]), float(var_range[1]))
    sd = OrderedDict()
    for name, (shape, dtype) in shapes.items():
        s = prng.name_seed(name, seed)
        if dtype == torch.int64:
            sd[name] = torch.tensor(0, dtype=torch.int64)
        elif name.endswith("bn.weight"):
            sd[name] = torch.from_numpy(prng.uniform(shape, s, 0.5, 1.5)) if calibrated else torch.ones(shape)
        elif name.endswith("bn.bias"):
            sd[name] = torch.from_numpy(prng.uniform(shape, s, -0.3, 0.3)) if calibrated else torch.zeros(shape)
        elif name.endswith("running_mean"):
            sd[name] = torch.from_numpy(prng.uniform(shape, s, -0.2, 0.2)) if calibrated else torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.from_numpy(prng.uniform(shape, s, v_lo, v_hi)) if calibrated else torch.ones(shape)
        elif name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / math.sqrt(fan_in)
            g = gain_cal if calibrated else 1.0   # 2.4 keeps activations O(1) through 17 ReLU layers
            sd[name] = torch.from_numpy(prng.uniform(shape, s, -b * g, b * g))
        elif name.endswith(".bias"):
            # fan_in of the matching weight
            wshape = shapes[name[:-4] + "weight"][0]
            b = 1.0 / math.sqrt(int(np.prod(wshape[1:])))
            sd[name] = torch.from_numpy(prng.uniform(shape, s, -b, b))
        else:
            raise KeyError(name)
    return sd


def synth_input(shape, seed):
    return torch.from_numpy(prng.uniform(shape, seed, 0.0, 1.0))


def disc_heatmaps(n, seq_len, h, w, seed, sigma=2.5):
    """Binary-disc targets (dataset.py:401-410 semantics):
    pixel (row i, col j) is 1 iff (i-cy)^2 + (j-cx)^2 <= sigma^2; every 5th map is empty."""
    r = prng.uniform((n, seq_len, 2), seed)
    y = np.zeros((n, seq_len, h, w), dtype=np.float32)
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    for a in range(n):
        for b in range(seq_len):
            if (a * seq_len + b) % 5 == 4:
                continue
            cx, cy = int(r[a, b, 0] * w), int(r[a, b, 1] * h)
            if cx == 0 and cy == 0:
                continue
            y[a, b] = ((ii - cy) ** 2 + (jj - cx) ** 2 <= sigma ** 2).astype(np.float32)
    return torch.from_numpy(y)


# --------------------------------------------------------------------------- TrackNet

def batchnorm2d(x, sd, prefix, training, stats_out=None):
    """nn.BatchNorm2d restated: model.py:9 (defaults eps=1e-5, momentum=0.1)."""
    g, b = sd[f"{prefix}.weight"].to(x.dtype), sd[f"{prefix}.bias"].to(x.dtype)
    if training:
        n = x.shape[0] * x.shape[2] * x.shape[3]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))      # biased
        if stats_out is not None:
            rm, rv = sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
            unbiased = var.detach() * (n / max(n - 1, 1))
            stats_out[f"{prefix}.running_mean"] = ((1 - BN_MOMENTUM) * rm.to(x.dtype) + BN_MOMENTUM * mean.detach())
            stats_out[f"{prefix}.running_var"] = ((1 - BN_MOMENTUM) * rv.to(x.dtype) + BN_MOMENTUM * unbiased)
            stats_out[f"{prefix}.num_batches_tracked"] = sd[f"{prefix}.num_batches_tracked"] + 1
    else:
        mean, var = sd[f"{prefix}.running_mean"].to(x.dtype), sd[f"{prefix}.running_var"].to(x.dtype)
    xhat = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return xhat * g[None, :, None, None] + b[None, :, None, None]


def conv2d_block(x, sd, prefix, training, stats_out=None, taps=None):
    z = F.conv2d(x, sd[f"{prefix}.conv.weight"].to(x.dtype), None, stride=1, padding=1)
    a = torch.relu(batchnorm2d(z, sd, f"{prefix}.bn", training, stats_out))
    if taps is not None:
        taps[prefix] = a
    return a


def _chain(x, sd, blk, n, training, stats_out, taps):
    for k in range(1, n + 1):
        x = conv2d_block(x, sd, f"{blk}.conv_{k}", training, stats_out, taps)
    return x


def upsample2x_nearest(x):
    """nn.Upsample(scale_factor=2) default mode 'nearest': out[i,j] = in[i//2, j//2]."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def tracknet_forward(sd, x, training=False, stats_out=None, taps=None, return_logits=False):
    """TrackNet.forward restated -- model.py:57-73."""
    x1 = _chain(x, sd, "down_block_1", 2, training, stats_out, taps)
    x = F.max_pool2d(x1, 2, 2)
    x2 = _chain(x, sd, "down_block_2", 2, training, stats_out, taps)
    x = F.max_pool2d(x2, 2, 2)
    x3 = _chain(x, sd, "down_block_3", 3, training, stats_out, taps)
    x = F.max_pool2d(x3, 2, 2)
    x = _chain(x, sd, "bottleneck", 3, training, stats_out, taps)
    x = torch.cat([upsample2x_nearest(x), x3], dim=1)
    x = _chain(x, sd, "up_block_1", 3, training, stats_out, taps)
    x = torch.cat([upsample2x_nearest(x), x2], dim=1)
    x = _chain(x, sd, "up_block_2", 2, training, stats_out, taps)
    x = torch.cat([upsample2x_nearest(x), x1], dim=1)
    x = _chain(x, sd, "up_block_3", 2, training, stats_out, taps)
    z = F.conv2d(x, sd["predictor.weight"].to(x.dtype), sd["predictor.bias"].to(x.dtype))
    if return_logits:
        return z
    return torch.sigmoid(z)


def wbce_loss(y_pred, y, reduce=True):
    """WBCELoss restated -- utils/metric.py:15-20."""
    p = y_pred
    loss = -(((1 - p) ** 2) * y * torch.log(torch.clamp(p, 1e-7, 1))
             + (p ** 2) * (1 - y) * torch.log(torch.clamp(1 - p, 1e-7, 1)))
    if reduce:
        return loss.mean()
    return loss.flatten(1).mean(1)


def wbce_grad_closed_form(p, y):
    """d(mean loss)/dp in closed form -- what the fused HIP kernel computes."""
    q = 1 - p
    pc, qc = torch.clamp(p, 1e-7, 1), torch.clamp(q, 1e-7, 1)
    in_p = ((p >= 1e-7) & (p <= 1)).to(p.dtype)
    in_q = ((q >= 1e-7) & (q <= 1)).to(p.dtype)
    g = -(-2 * q * y * torch.log(pc) + q * q * y * in_p / pc
          + 2 * p * (1 - y) * torch.log(qc) - p * p * (1 - y) * in_q / qc)
    return g / p.numel()


def mixup_injected(x, y, lamb, index):
    """mixup with injected lambda / permutation -- train.py:32-40 (RNG is not part of parity)."""
    lamb = np.maximum(lamb, 1 - lamb)
    lam = torch.from_numpy(np.asarray(lamb)[:, None, None, None]).float().to(x.dtype)
    index = torch.as_tensor(index, dtype=torch.long)
    return x * lam + x[index] * (1 - lam), y * lam + y[index] * (1 - lam)


def tracknet_train_step_grads(sd, x, y, dtype=torch.float32):
    """One forward(train mode)+WBCE+backward; returns loss, heatmap, grads, new BN buffers."""
    names = [k for k, v in sd.items() if v.dtype != torch.int64 and "running_" not in k]
    work = OrderedDict((k, (v.to(dtype) if v.dtype != torch.int64 else v)) for k, v in sd.items())
    for k in names:
        work[k] = work[k].clone().requires_grad_(True)
    stats = {}
    p = tracknet_forward(work, x.to(dtype), training=True, stats_out=stats)
    loss = wbce_loss(p, y.to(dtype))
    grads = torch.autograd.grad(loss, [work[k] for k in names])
    return loss.detach(), p.detach(), OrderedDict(zip(names, grads)), stats


# --------------------------------------------------------------------------- InpaintNet

def _conv1d_block(x, sd, prefix):
    z = F.conv1d(x, sd[f"{prefix}.weight"].to(x.dtype), sd[f"{prefix}.bias"].to(x.dtype), padding=1)
    return F.leaky_relu(z, 0.01)


def inpaintnet_forward(sd, x, m):
    """InpaintNet.forward restated -- model.py:113-129.  x (N,L,2), m (N,L,1) -> (N,L,2)."""
    x = torch.cat([x, m.to(x.dtype)], dim=2).permute(0, 2, 1)
    x1 = _conv1d_block(x, sd, "down_1.conv")
    x2 = _conv1d_block(x1, sd, "down_2.conv")
    x3 = _conv1d_block(x2, sd, "down_3.conv")
    x = _conv1d_block(x3, sd, "buttleneck.conv_1.conv")
    x = _conv1d_block(x, sd, "buttleneck.conv_2.conv")
    x = _conv1d_block(torch.cat([x, x3], dim=1), sd, "up_1.conv")
    x = _conv1d_block(torch.cat([x, x2], dim=1), sd, "up_2.conv")
    x = _conv1d_block(torch.cat([x, x1], dim=1), sd, "up_3.conv")
    z = F.conv1d(x, sd["predictor.weight"].to(x.dtype), sd["predictor.bias"].to(x.dtype), padding=1)
    return torch.sigmoid(z).permute(0, 2, 1)


def inpaint_masked_mse(refine, gt, mask):
    """train.py:159-161 -- nn.MSELoss()(refine*mask, gt*mask): mean over ALL N*L*2 elements."""
    return (((refine - gt) * mask) ** 2).mean()
