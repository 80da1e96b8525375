"""TrackNet / InpaintNet with the reference's module tree and ``state_dict`` layout, computed by libtnv3_hip.so.

Drop-in for the reference's ``model.py`` (TrackNet: model.py:45-73, InpaintNet: model.py:100-129): same
constructor arguments, same attribute names (including the misspelt ``buttleneck``), therefore the same 104 / 18
``state_dict`` keys, shapes and dtypes, so reference checkpoints load with
``load_state_dict(ckpt['model'])`` and ``torch.optim`` / ``clip_grad_norm_`` see ordinary leaf Parameters.

The sub-modules below are parameter containers only: they never run a torch convolution.  ``forward`` enqueues
hand-written gfx950 kernels through the C ABI (``ops``).  There is no CPU or ATen fallback: on a non-GPU tensor, or
without the HIP library, the call raises.
"""
import contextlib
import math
import threading

import torch
import torch.nn as nn

from . import ops
from . import tuning


# --------------------------------------------------------------------------- parameter containers
class _Conv3x3Params(nn.Module):
    """Holds the filter of ``nn.Conv2d(in, out, 3, padding='same', bias=False)`` (model.py:8); default PyTorch init."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim, 3, 3))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class _BatchNormParams(nn.Module):
    """Holds the state of ``nn.BatchNorm2d(out)`` (model.py:9): weight, bias, running stats, step counter."""

    def __init__(self, dim, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.register_buffer("running_mean", torch.zeros(dim))
        self.register_buffer("running_var", torch.ones(dim))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Conv2DBlock(nn.Module):
    """Conv3x3 + BN + ReLU (model.py:4-16) as ONE fused kernel in eval mode."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv = _Conv3x3Params(in_dim, out_dim)
        self.bn = _BatchNormParams(out_dim)
        self._cache = {}
        self._wino_plan = set()          # Winograd panels requested so far: what repack_wino_panels() rebuilds in one launch ...
        self._wino_used = set()          # ... if they were asked for since the last repack (an eval-only panel is not rebuilt per training step)

    def _versions(self, names):
        out = []
        for n in names:
            t = getattr(self.bn, n) if n != "conv" else self.conv.weight
            out.append((t.data_ptr(), t._version))
        return tuple(out)

    def packed_weight(self, transpose_flip=False):
        key = ("wd" if transpose_flip else "wf")
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_conv3x3_weights(self.conv.weight.detach(), transpose_flip=transpose_flip))
            self._cache[key] = hit
        return hit[1]

    def packed_wino(self, c_from=0):
        """Winograd-domain filters G w G^T of input channels c_from.. (the whole layer, or the skip half of a decoder entry)."""
        # the panel layout follows the kernel variant in use, which follows the channel counts of the (sub-)layer
        key = ("wino", int(c_from), ops.wino_layout(None, self.conv.in_dim - int(c_from), self.conv.out_dim))
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = self.conv.weight.detach()
            hit = (ver, ops.pack_wino_weights(w, c_from=c_from))
            self._cache[key] = hit
            self._wino_plan.add((key, int(c_from), False))
        self._wino_used.add((key, int(c_from), False))
        return hit[1]

    def packed_wino_t(self, c_from=0):
        """Winograd-domain filters of the data gradient: G w' G^T with w'[ci][co][kh][kw] = w[co][c_from + ci][2-kh][2-kw]."""
        key = ("wino_t", int(c_from), ops.wino_layout(None, self.conv.out_dim, self.conv.in_dim - int(c_from)))      # dgrad: Cout -> c_count channels
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = self.conv.weight.detach()
            hit = (ver, ops.pack_wino_weights(w, c_from=c_from, transpose_flip=True))
            self._cache[key] = hit
            self._wino_plan.add((key, int(c_from), True))
        self._wino_used.add((key, int(c_from), True))
        return hit[1]

    def packed_wino43(self, c_from=0):
        """Winograd F(4x4, 3x3) filter panel of input channels c_from.. (ops.conv3x3_wino43)."""
        key = ("w43", int(c_from), ops.wino43_variant(None))       # (the panel layout follows the kernel variant)
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_wino43_weights(self.conv.weight.detach(), c_from=c_from, variant=key[2]))
            self._cache[key] = hit
            self._wino_plan.add((key, int(c_from), False))
        self._wino_used.add((key, int(c_from), False))
        return hit[1]

    def packed_wino43_t(self, c_from=0):
        """F(4x4, 3x3) panel of the data gradient's filter w'[ci][co][kh][kw] = w[co][c_from + ci][2-kh][2-kw]."""
        key = ("w43t", int(c_from), ops.wino43_variant(None))
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_wino43_weights(self.conv.weight.detach(), c_from=c_from, transpose_flip=True, variant=key[2]))
            self._cache[key] = hit
            self._wino_plan.add((key, int(c_from), True))
        self._wino_used.add((key, int(c_from), True))
        return hit[1]

    def stale_wino_panels(self):
        """[(cache key, c_from, transpose_flip)] of the Winograd panels this block has been asked for (by the forward / backward paths
        actually taken: the plan follows the shapes and tuning switches by construction) since the last one-launch repack -- a panel only
        the eval forward reads is not rebuilt in front of every training step; it is repacked lazily by the next eval forward -- whose
        weight has changed since."""
        if not self._wino_plan:
            return []
        ver = self._versions(["conv"])
        return [it for it in self._wino_plan if it in self._wino_used and self._cache.get(it[0], (None,))[0] != ver]

    def packed_up2x(self, c0):
        """(class filters of the first c0 = upsampled input channels, packed 3x3 filter of the remaining skip channels):
        the two operands of the decoder-entry formulation (ops.conv_up2x + ops.conv3x3(..., addend=...))."""
        key = ("up2x", int(c0))
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = self.conv.weight.detach()
            hit = (ver, (ops.pack_up2x_weights(w, c0), ops.pack_conv3x3_weights(w[:, c0:].contiguous())))
            self._cache[key] = hit
        return hit[1]

    def packed_up2x_wino(self, c0, variant=None):
        """U' of the first c0 (upsampled) input channels for the Winograd form of the low-resolution half (ops.conv_up2x_wino; the panel
        follows the kernel variant)."""
        v = 2 if ops.up2x_wino_variant(variant) == 2 else 0
        key = ("up2xw", int(c0), v)
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_up2x_wino_weights(self.conv.weight.detach(), c0, variant=v))
            self._cache[key] = hit
        return hit[1]

    def packed_dgrad_up2x_wino(self, c0, variant=None):
        """U'' of the first c0 (upsampled) input channels for the low-resolution data gradient (ops.dgrad_up2x_wino; the panel follows the
        kernel variant)."""
        v = 2 if ops.dgrad_up2x_wino_variant(variant) == 2 else 0
        key = ("dup2xw", int(c0), v)
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_dgrad_up2x_wino_weights(self.conv.weight.detach(), c0, variant=v))
            self._cache[key] = hit
        return hit[1]

    def packed_dgrad_up2x(self, c0):
        """(4x4 stride-2 filters of the low-resolution data gradient w.r.t. the first c0 inputs, transposed / flipped 3x3
        filter of the remaining skip channels): the backward twins of packed_up2x."""
        key = ("dup2x", int(c0))
        ver = self._versions(["conv"])
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = self.conv.weight.detach()
            hit = (ver, (ops.pack_dgrad_up2x_weights(w, c0), ops.pack_conv3x3_weights(w[:, c0:].contiguous(), transpose_flip=True)))
            self._cache[key] = hit
        return hit[1]

    def conv_up_skip(self, x_low, skip, n, relu, affine, want_stats=False, layer=None):
        """conv3x3(cat([upsample2x(x_low), skip], 1)) with the upsampled half computed at the low resolution.  want_stats (training
        forward, Winograd skip half): returns (z, tile_stats) -- BatchNorm's batch statistics from the kernel's epilogue."""
        c0, c1 = int(x_low.shape[1]), int(skip.shape[1])
        h, w = int(skip.shape[2]), int(skip.shape[3])
        uv = ops.up2x_wino_variant(tuning.UP2X_WINO_VARIANT if affine else tuning.UP2X_WINO_VARIANT_TRAIN)
        if tuning.UP2X_WINO and not ops.up2x_wino_supported(c0, self.conv.out_dim, h // 2, w // 2, uv):
            uv = 0
        if tuning.UP2X_WINO and ops.up2x_wino_supported(c0, self.conv.out_dim, h // 2, w // 2, uv):
            part = ops.conv_up2x_wino(x_low, self.packed_up2x_wino(c0, uv), self.conv.out_dim, variant=uv)     # 9 of the 16 F(2x2) GEMMs / 25 of the 36 F(4x4) products
            wskip = None
        else:
            wq, wskip = self.packed_up2x(c0)
            part = ops.conv_up2x(x_low, wq, self.conv.out_dim)
        if wskip is None and not (c1 >= tuning.WINOGRAD_MIN_SKIP and tuning.use_winograd(c1, self.conv.out_dim, h, w)):
            wskip = self.packed_up2x(c0)[1]
        cfg = tuning.conv_config(self.conv.out_dim, c1, n, h, w)
        bn = self.bn
        if affine and c1 >= tuning.WINOGRAD_MIN_SKIP and tuning.use_wino43(c1, self.conv.out_dim, h, w):     # eval mode: the skip half in F(4x4, 3x3) form
            return ops.conv3x3_wino43(skip, self.packed_wino43(c0), self.conv.out_dim, mean=bn.running_mean, scale=self.eval_scale(),
                                      shift=bn.bias.detach(), relu=relu, addend=part)
        if affine and c1 >= tuning.WINOGRAD_MIN_SKIP and tuning.use_winograd(c1, self.conv.out_dim, h, w):   # eval mode: the skip half in Winograd form
            return ops.conv3x3_wino(skip, self.packed_wino(c0), self.conv.out_dim, mean=bn.running_mean, scale=self.eval_scale(),
                                    shift=bn.bias.detach(), relu=relu, addend=part)
        if affine:
            return ops.conv3x3(skip, wskip, self.conv.out_dim, addend=part, mean=bn.running_mean, scale=self.eval_scale(),
                               shift=bn.bias.detach(), relu=relu, cfg=cfg)
        if want_stats and c1 >= tuning.WINOGRAD_MIN_SKIP and tuning.use_wino43_train(c1, self.conv.out_dim, h, w, layer=layer):   # training forward, F(4x4, 3x3)
            return ops.conv3x3_wino43_stats(skip, self.packed_wino43(c0), self.conv.out_dim, addend=part)
        if c1 >= tuning.WINOGRAD_MIN_SKIP and tuning.use_winograd(c1, self.conv.out_dim, h, w):   # training forward: raw sums
            if want_stats and tuning.wino_has_stats():
                return ops.conv3x3_wino_stats(skip, self.packed_wino(c0), self.conv.out_dim, addend=part)
            z = ops.conv3x3_wino(skip, self.packed_wino(c0), self.conv.out_dim, addend=part, relu=relu)
            return (z, None) if want_stats else z
        z = ops.conv3x3(skip, wskip, self.conv.out_dim, addend=part, relu=relu, cfg=cfg)
        return (z, None) if want_stats else z

    def invalidate_caches(self):
        """Drop every cached operand (packed / Winograd-domain filters, folded BN scale).  The caches are keyed on the tensors'
        version counters, which ordinary in-place ops (optimiser steps, load_state_dict, copy_ / mul_ on the tensor or on
        tensor.detach()) bump -- writes through `.data` or through raw pointers do NOT, and need this call."""
        self._cache.clear()

    def eval_scale(self):
        """gamma / sqrt(running_var + eps), recomputed when gamma or running_var change (running_var is also written by the
        training-mode BN kernel through a raw pointer, which no version counter sees: num_batches_tracked, bumped with
        every training forward, is part of the key, and the training forward drops the entry as well)."""
        ver = self._versions(["weight", "running_var", "num_batches_tracked"])
        hit = self._cache.get("aff")
        if hit is None or hit[0] != ver:
            bn = self.bn
            hit = (ver, ops.bn_eval_scale(bn.weight.detach(), bn.running_var, bn.eps))
            self._cache["aff"] = hit
        return hit[1]

    def forward_eval(self, x, skip=None, up=False, pool=False):
        """pool: returns (y, maxpool2x2(y)) -- from the F(4x4) kernel's write-out where that kernel runs the layer, else a separate pass."""
        if pool:
            h, w = int(x.shape[2]), int(x.shape[3])
            if skip is None and not up and tuning.FUSE_POOL and tuning.use_wino43(self.conv.in_dim, self.conv.out_dim, h, w):
                bn = self.bn
                return ops.conv3x3_wino43(x, self.packed_wino43(), self.conv.out_dim, mean=bn.running_mean, scale=self.eval_scale(),
                                          shift=bn.bias.detach(), relu=True, pool=True)
            y = self.forward_eval(x, skip=skip, up=up)
            return y, ops.maxpool2x2(y)
        bn = self.bn
        n = x.shape[0]
        if up and skip is not None:
            return self.conv_up_skip(x, skip, int(n), relu=True, affine=True)
        h = x.shape[2] * (2 if up else 1)
        w = x.shape[3] * (2 if up else 1)
        if skip is None and not up and tuning.use_wino43(self.conv.in_dim, self.conv.out_dim, int(h), int(w)):
            return ops.conv3x3_wino43(x, self.packed_wino43(), self.conv.out_dim, mean=bn.running_mean, scale=self.eval_scale(),
                                      shift=bn.bias.detach(), relu=True)
        if skip is None and not up and tuning.use_winograd(self.conv.in_dim, self.conv.out_dim, int(h), int(w)):
            return ops.conv3x3_wino(x, self.packed_wino(), self.conv.out_dim, mean=bn.running_mean, scale=self.eval_scale(),
                                    shift=bn.bias.detach(), relu=True)
        cfg = tuning.conv_config(self.conv.out_dim, self.conv.in_dim, n, h, w)
        return ops.conv3x3(x, self.packed_weight(), self.conv.out_dim, src1=skip, mean=bn.running_mean, scale=self.eval_scale(),
                           shift=bn.bias.detach(), up0=up, relu=True, cfg=cfg)

    def forward(self, x):
        if self.training:
            raise RuntimeError("a Conv2DBlock is trained through TrackNet.forward (the whole network is one autograd node); "
                               "call the block directly only in eval mode")
        with torch.no_grad():
            return self.forward_eval(x.contiguous())


class Double2DConv(nn.Module):
    """Conv2DBlock x 2 (model.py:18-28)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv_1 = Conv2DBlock(in_dim, out_dim)
        self.conv_2 = Conv2DBlock(out_dim, out_dim)

    def blocks(self):
        return [self.conv_1, self.conv_2]


class Triple2DConv(nn.Module):
    """Conv2DBlock x 3 (model.py:30-42)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv_1 = Conv2DBlock(in_dim, out_dim)
        self.conv_2 = Conv2DBlock(out_dim, out_dim)
        self.conv_3 = Conv2DBlock(out_dim, out_dim)

    def blocks(self):
        return [self.conv_1, self.conv_2, self.conv_3]


class _Conv1x1Params(nn.Module):
    """Holds ``nn.Conv2d(64, out_dim, (1, 1))`` (model.py:54): weight (L,64,1,1) + bias (L,), default init."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim, 1, 1))
        self.bias = nn.Parameter(torch.empty(out_dim))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_dim)
        nn.init.uniform_(self.bias, -bound, bound)


class TrackNet(nn.Module):
    """VGG-style U-Net heat-map network (model.py:45-73).  in: (N, in_dim, H, W) -> out: (N, out_dim, H, W)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.down_block_1 = Double2DConv(in_dim, 64)
        self.down_block_2 = Double2DConv(64, 128)
        self.down_block_3 = Triple2DConv(128, 256)
        self.bottleneck = Triple2DConv(256, 512)
        self.up_block_1 = Triple2DConv(768, 256)
        self.up_block_2 = Double2DConv(384, 128)
        self.up_block_3 = Double2DConv(192, 64)
        self.predictor = _Conv1x1Params(64, out_dim)

    # -- eval: 17 fused conv+BN+ReLU kernels, 3 pools, 1 head; upsample+concat folded into the consumer's loader
    @staticmethod
    def _chain_eval(blocks, x, skip=None, up=False, pool=False):
        """pool: (y, maxpool2x2(y)) of the chain's last layer."""
        x = blocks[0].forward_eval(x, skip=skip, up=up)
        for b in blocks[1:-1]:
            x = b.forward_eval(x)
        return blocks[-1].forward_eval(x, pool=pool)

    def _forward_eval(self, x, out=None):
        """out: where the head writes the heat maps (a contiguous (n, out_dim, H, W) view, e.g. a batch slice of the caller's tensor)."""
        x1, p1 = self._chain_eval(self.down_block_1.blocks(), x, pool=True)
        x2, p2 = self._chain_eval(self.down_block_2.blocks(), p1, pool=True)
        x3, p3 = self._chain_eval(self.down_block_3.blocks(), p2, pool=True)
        x = self._chain_eval(self.bottleneck.blocks(), p3)
        x = self._chain_eval(self.up_block_1.blocks(), x, skip=x3, up=True)
        x = self._chain_eval(self.up_block_2.blocks(), x, skip=x2, up=True)
        x = self._chain_eval(self.up_block_3.blocks(), x, skip=x1, up=True)
        return ops.head1x1_sigmoid(x, self.predictor.weight.detach(), self.predictor.bias.detach(), out=out)

    def invalidate_caches(self):
        """Drop the cached packed operands of every block (needed after writes through `.data` / raw pointers, which bump no
        version counter; ordinary in-place updates are tracked automatically)."""
        for m in self.modules():
            if isinstance(m, Conv2DBlock):
                m.invalidate_caches()

    def repack_wino_panels(self):
        """Rebuild, in ONE launch on the current stream, every Winograd filter panel (forward and data-gradient) the blocks have been
        asked for before and whose weights changed since -- what a training step does after each optimiser step, otherwise as ~33
        14-us launches strung along the main stream.  Purely an accelerator of the lazy per-block caches: a panel that is not in a
        block's plan yet (first step, new shape) is packed by the call that needs it."""
        todo = []
        for m in self.modules():
            if isinstance(m, Conv2DBlock):
                for it in m.stale_wino_panels():
                    todo.append((m, it))
        if len(todo) < 2:
            return 0
        with torch.no_grad():
            # (the layout a panel is rebuilt in is the one its cache key recorded, not what the tuning switches say now)
            panels = ops.pack_wino_weights_multi([(m.conv.weight.detach(), c_from, flip, 43 if key[0] in ("w43", "w43t") else 22,
                                                   (3 if key[2] == 1 else 4) if key[0] in ("w43", "w43t") else key[2])
                                                  for m, (key, c_from, flip) in todo])
        for (m, (key, _, _)), u in zip(todo, panels):
            m._cache[key] = (m._versions(["conv"]), u)
        for m in self.modules():
            if isinstance(m, Conv2DBlock):
                m._wino_used.clear()
        return len(todo)

    def prepare_eval(self):
        """Build (on the current stream) every cached eval-mode operand -- packed filters, folded BN scales -- so that
        forwards issued afterwards on several streams only read them."""
        with torch.no_grad():
            for blk in (self.down_block_1, self.down_block_2, self.down_block_3, self.bottleneck, self.up_block_1,
                        self.up_block_2, self.up_block_3):
                for i, b in enumerate(blk.blocks()):
                    if i == 0 and blk in (self.up_block_1, self.up_block_2, self.up_block_3):
                        c0 = b.conv.in_dim * 2 // 3               # decoder entry: 2/3 of the inputs are the upsampled tensor
                        b.packed_up2x(c0)
                        b.packed_up2x_wino(c0)
                        b.packed_wino(c0)
                        if tuning.WINOGRAD and tuning.WINO43 and b.conv.out_dim % 64 == 0 and b.conv.in_dim - c0 >= tuning.WINO43_MIN_CIN:
                            b.packed_wino43(c0)
                    else:
                        b.packed_weight()
                        if b.conv.in_dim >= tuning.WINOGRAD_MIN_CIN:
                            b.packed_wino()
                        if tuning.WINOGRAD and tuning.WINO43 and b.conv.out_dim % 64 == 0 and b.conv.in_dim >= tuning.WINO43_MIN_CIN:
                            b.packed_wino43()
                    b.eval_scale()

    def _forward_eval_split(self, x):
        """The eval forward with the batch split over the current stream and side streams (tuning.INFER_SPLIT_PARTS, default 6 : 4 over
        two).  Images are independent in eval mode, so the outputs are those of _forward_eval to the last bit; the parts' per-layer
        launches overlap at their tails, where a single launch leaves CUs idle."""
        dev = x.device
        n = int(x.shape[0])
        parts = tuning.INFER_SPLIT_PARTS
        tot = sum(parts)
        cuts, acc = [0], 0
        for p in parts[:-1]:                                 # cumulative rounding: 6 : 4 of 10 -> 6 + 4, of 16 -> 10 + 6
            acc += p
            cuts.append(min(n, (acc * n + tot // 2) // tot))
        cuts.append(n)
        main = torch.cuda.current_stream(dev)
        self.prepare_eval()                                  # cached operands are built on this stream, before the side streams read them
        ready = torch.cuda.Event()
        ready.record(main)                                   # x (and the caches) were produced on this stream
        # one output tensor, allocated on this stream; every part's head writes its batch slice (no concatenation pass)
        y = torch.empty((n, self.out_dim, int(x.shape[2]), int(x.shape[3])), dtype=torch.float32, device=dev)
        sides = []
        for i in range(1, len(parts)):
            if cuts[i + 1] <= cuts[i]:
                continue
            side = _split_stream(dev, i - 1)
            sides.append(side)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                self._forward_eval(x[cuts[i]:cuts[i + 1]], out=y[cuts[i]:cuts[i + 1]])
        self._forward_eval(x[:cuts[1]], out=y[:cuts[1]])
        for side in sides:
            main.wait_stream(side)                           # (y lives on this stream and is handed on only after the join)
        return y

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_dim:
            raise ValueError(f"TrackNet expects (N, {self.in_dim}, H, W), got {tuple(x.shape)}")
        if (x.shape[2] % 8) or (x.shape[3] % 8):
            raise ValueError("TrackNet needs H and W divisible by 8 (three 2x2 poolings)")
        x = x.contiguous()
        if self.training:
            from . import autograd_ops
            return autograd_ops.tracknet_forward_train(self, x)
        with torch.no_grad():
            n = int(x.shape[0])
            if (x.is_cuda and tuning.INFER_SPLIT and not _NO_SPLIT.active and n >= tuning.INFER_SPLIT_MIN_BATCH
                    and n * int(x.shape[2]) * int(x.shape[3]) >= tuning.INFER_SPLIT_MIN_PIXELS and not torch.cuda.is_current_stream_capturing()):
                return self._forward_eval_split(x)
            return self._forward_eval(x)


_SPLIT_STREAMS = {}


def _split_stream(dev, i=0):
    """The i-th side stream of the intra-batch split, per device (the caching allocator pools memory per stream: a fresh stream per
    call would pay a hipMalloc for every activation)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), i)
    if key not in _SPLIT_STREAMS:
        _SPLIT_STREAMS[key] = torch.cuda.Stream(dev)
    return _SPLIT_STREAMS[key]


class _NoSplit(threading.local):
    active = False


_NO_SPLIT = _NoSplit()


@contextlib.contextmanager
def no_infer_split():
    """Callers that already keep several batches in flight on their own streams (pipeline.predict_video) switch the intra-batch
    split off for their forwards."""
    prev, _NO_SPLIT.active = _NO_SPLIT.active, True
    try:
        yield
    finally:
        _NO_SPLIT.active = prev


# --------------------------------------------------------------------------- InpaintNet
class _Conv1dParams(nn.Module):
    """Holds ``nn.Conv1d(in, out, kernel_size=3, padding='same', bias=True)`` (model.py:80,110); default init."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim, 3))
        self.bias = nn.Parameter(torch.empty(out_dim))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_dim * 3)
        nn.init.uniform_(self.bias, -bound, bound)


class Conv1DBlock(nn.Module):
    """Conv1D + LeakyReLU (model.py:76-87); container only."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv = _Conv1dParams(in_dim, out_dim)


class Double1DConv(nn.Module):
    """Conv1DBlock x 2 (model.py:89-98)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv_1 = Conv1DBlock(in_dim, out_dim)
        self.conv_2 = Conv1DBlock(out_dim, out_dim)


class InpaintNet(nn.Module):
    """1-D trajectory rectification network (model.py:100-129).  x (N,L,2), m (N,L,1) -> (N,L,2)."""

    def __init__(self):
        super().__init__()
        self.down_1 = Conv1DBlock(3, 32)
        self.down_2 = Conv1DBlock(32, 64)
        self.down_3 = Conv1DBlock(64, 128)
        self.buttleneck = Double1DConv(128, 256)
        self.up_1 = Conv1DBlock(384, 128)
        self.up_2 = Conv1DBlock(192, 64)
        self.up_3 = Conv1DBlock(96, 32)
        self.predictor = _Conv1dParams(32, 2)

    _TOPS = ("down_1", "down_2", "down_3", "buttleneck", "up_1", "up_2", "up_3", "predictor")

    def conv_params(self):
        """[(weight, bias)] in forward order.  (The nine leaf modules are looked up once: nn.Module attribute access costs ~1 us a
        piece and this runs on every forward -- 24 of the 53 us a batch-32 forward took end to end; the cache is dropped when one of
        the eight top-level children is replaced.)"""
        mods = self._modules
        hit = self.__dict__.get("_conv_leaves")
        if hit is None or any(mods[k] is not t for k, t in zip(self._TOPS, hit[0])):
            leaves = [self.down_1.conv, self.down_2.conv, self.down_3.conv, self.buttleneck.conv_1.conv,
                      self.buttleneck.conv_2.conv, self.up_1.conv, self.up_2.conv, self.up_3.conv, self.predictor]
            hit = (tuple(mods[k] for k in self._TOPS), leaves)
            self.__dict__["_conv_leaves"] = hit
        return [(m._parameters["weight"], m._parameters["bias"]) for m in hit[1]]

    def forward(self, x, m):
        if x.dim() != 3 or x.shape[2] != 2 or m.shape[:2] != x.shape[:2] or m.shape[2] != 1:
            raise ValueError(f"InpaintNet expects x (N,L,2) and m (N,L,1), got {tuple(x.shape)} / {tuple(m.shape)}")
        from . import inpaint_ops
        return inpaint_ops.inpaintnet_forward(self, x, m)
