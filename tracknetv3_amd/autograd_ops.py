"""Training-mode forward/backward of TrackNet as ONE torch.autograd.Function over HIP kernels.

The reference relies on PyTorch autograd through ~60 ATen ops (train.py:92-95).  Here the whole network is a single
autograd node: ``forward`` runs conv (fp32 MFMA) -> BN batch-stats -> normalise+ReLU per Conv2DBlock and keeps what the
backward needs (raw conv outputs z, activations a, saved mean/invstd); ``backward`` receives dL/dp from the loss
node (``WBCELoss`` below, or any torch op) and runs head-backward, then per block in reverse: BN+ReLU backward ->
data gradient (the forward MFMA kernel on transposed/flipped filters) -> weight gradient (MFMA split-K), with the
pool / upsample / concat gradients folded into three small kernels.  Parameters stay ordinary leaf tensors, so
``torch.optim`` and ``state_dict`` work unchanged.  Optional ``grad_ready`` hook: called with (param, grad) as soon as
a gradient is final, which is what the data-parallel wrapper uses to overlap RCCL all-reduce with the rest of backward.
"""
import os

import torch

from . import ops
from . import tuning

_grad_ready_hook = None
_backward_end_hook = None
_grad_dest_hook = None

# Weight gradients on a side HIP stream: wgrad(L) depends only on dZ(L), while the main chain continues with
# dgrad(L) -> BN backward(L-1) -> ...  Two independent chains in flight let the HBM-bound BN passes and the tails of
# the batch-10 conv launches hide under the other chain's MFMA work.
_wgrad_overlap = os.environ.get("TNV3_WGRAD_OVERLAP", "1") != "0"
_WGRAD_STREAMS = {}


def set_wgrad_overlap(enabled):
    """Enable / disable the side-stream weight gradients (default on for GPU tensors); returns the previous setting."""
    global _wgrad_overlap
    old, _wgrad_overlap = _wgrad_overlap, bool(enabled)
    return old


_BACKWARD_STREAMS = {}


def backward_streams(dev):
    """Streams the running (or last) backward on `dev` produces gradients on: [main] or [main, weight-gradient stream]."""
    return _BACKWARD_STREAMS.get(dev.index if dev.index is not None else 0, [])


def wgrad_stream(dev):
    """The persistent side stream weight gradients are launched on (None when the overlap is off or `dev` is no GPU)."""
    if not _wgrad_overlap or dev.type != "cuda":
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(dev, priority=tuning.WGRAD_STREAM_PRIORITY)
    return _WGRAD_STREAMS[key]


def set_grad_ready_hook(fn, end_fn=None, dest_fn=None):
    """fn(param, grad) -> grad-or-replacement is called inside backward as soon as `grad` is final; end_fn() runs
    once after the last gradient of the step (None disables both).  dest_fn(param) -> tensor-or-None names where the kernel that
    produces `param`'s gradient should WRITE it (the data-parallel reducer returns the parameter's view of its flat all-reduce bucket:
    the gradient then needs no copy into the bucket); fn still sees every gradient."""
    global _grad_ready_hook, _backward_end_hook, _grad_dest_hook
    _grad_ready_hook, _backward_end_hook, _grad_dest_hook = fn, end_fn, dest_fn


def grad_ready_order(net):
    """Parameters in the order their gradients become final inside backward (head first, first conv last)."""
    order = [net.predictor.weight, net.predictor.bias]
    for blks in reversed(_blocks(net)):
        for blk in reversed(blks):
            order += [blk.bn.weight, blk.bn.bias, blk.conv.weight]
    return order


def _blocks(net):
    return (net.down_block_1.blocks(), net.down_block_2.blocks(), net.down_block_3.blocks(), net.bottleneck.blocks(),
            net.up_block_1.blocks(), net.up_block_2.blocks(), net.up_block_3.blocks())


def _cfg(blk, n, h, w):
    return tuning.conv_config(blk.conv.out_dim, blk.conv.in_dim, n, h, w)


def _train_forward_body(net, x):
    """conv -> BN(batch statistics) -> ReLU per Conv2DBlock up to the head's input; returns (saved records, head input, skips)."""
    saved = []          # per block: dict(x0, x1, up, z, a, mean, invstd)
    bumped = []
    n = x.shape[0]
    if tuning.WINO_REPACK_MULTI:
        net.repack_wino_panels()        # the panels the optimiser step made stale, forward and backward ones, in one launch

    pooled_of = {}         # id(a) -> MaxPool2d(2, 2)(a), written by the BatchNorm + ReLU pass of a down block's last layer

    def block_fwd(blk, src0, src1=None, up=False, pool=False):
        h = src0.shape[2] * (2 if up else 1)
        w = src0.shape[3] * (2 if up else 1)
        stats = None           # BatchNorm's batch statistics taken in the conv epilogue (Winograd kernels 3 / 4), else a pass over z
        if up and src1 is not None:
            z, stats = blk.conv_up_skip(src0, src1, int(n), relu=False, affine=False, want_stats=True, layer=len(saved))
        elif src1 is None and not up and tuning.use_wino43_train(blk.conv.in_dim, blk.conv.out_dim, int(h), int(w), layer=len(saved)):
            z, stats = ops.conv3x3_wino43_stats(src0, blk.packed_wino43(), blk.conv.out_dim)      # F(4x4, 3x3) + statistics epilogue
        elif src1 is None and not up and tuning.use_winograd(blk.conv.in_dim, blk.conv.out_dim, int(h), int(w)):
            if tuning.wino_has_stats():
                z, stats = ops.conv3x3_wino_stats(src0, blk.packed_wino(), blk.conv.out_dim)
            else:
                z = ops.conv3x3_wino(src0, blk.packed_wino(), blk.conv.out_dim)      # raw conv output in Winograd form
        else:
            z = ops.conv3x3(src0, blk.packed_weight(), blk.conv.out_dim, src1=src1, up0=up, relu=False, cfg=_cfg(blk, n, h, w))
        bn = blk.bn
        if pool and tuning.POOL_IN_BN_APPLY:
            a, mean, invstd, pooled_of["y"] = ops.bn_train_forward(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                                   bn.eps, bn.momentum, tile_stats=stats, pool=True)
        else:
            a, mean, invstd = ops.bn_train_forward(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                                   bn.eps, bn.momentum, tile_stats=stats)
            if pool:
                pooled_of["y"] = ops.maxpool2x2(a)
        bumped.append(bn.num_batches_tracked)           # (+= 1 for all 17 layers in ONE launch behind the last layer: they were 17 launches on the chain)
        blk._cache.pop("aff", None)       # running_var was rewritten through a raw pointer: the folded eval scale is stale
        saved.append(dict(blk=blk, idx=len(saved), x0=src0, x1=src1, up=up, z=z, a=a, mean=mean, invstd=invstd,
                          bn_ver=(bn.weight._version, bn.bias._version)))
        return a

    def chain(blks, src0, src1=None, up=False, pool=False):
        """pool: the chain's output is pooled next -- returns (y, MaxPool2d(2, 2)(y))"""
        y = block_fwd(blks[0], src0, src1, up, pool=pool and len(blks) == 1)
        for k, b in enumerate(blks[1:]):
            y = block_fwd(b, y, pool=pool and k == len(blks) - 2)
        return (y, pooled_of.pop("y")) if pool else y

    d1, d2, d3, bt, u1, u2, u3 = _blocks(net)
    x1, p1 = chain(d1, x, pool=True)
    x2, p2 = chain(d2, p1, pool=True)
    x3, p3 = chain(d3, p2, pool=True)
    y = chain(bt, p3)
    y = chain(u1, y, x3, True)
    y = chain(u2, y, x2, True)
    y = chain(u3, y, x1, True)
    torch._foreach_add_(bumped, 1)
    return saved, y, (x1, x2, x3)


def _train_backward_body(ctx, dev, head_backward):
    """Everything behind the loss: `head_backward((dW destination, db destination))` -> (da, dW_head, db_head), then per block in reverse BN+ReLU backward,
    data gradient, weight gradient (side stream).  Returns the tuple autograd expects after the non-tensor arguments."""
    net, saved = ctx.net, ctx.saved
    if saved is None:
        raise RuntimeError("TrackNet backward ran twice over the same forward: the training node frees its activations at the "
                           "end of backward (retain_graph=True is not supported); run the forward again")
    hook = _grad_ready_hook
    dest_fn = _grad_dest_hook
    grads = {}
    keep = []

    def dest(param):
        return dest_fn(param) if dest_fn is not None else None

    def done(param, g):
        if hook is not None:
            r = hook(param, g)
            if r is not None:
                g = r
        grads[id(param)] = g

    side = wgrad_stream(dev)
    main = torch.cuda.current_stream(dev) if side is not None else None
    side_results = []                                      # [(tensor made on the side stream, its completion event)]
    if dev.type == "cuda":
        _BACKWARD_STREAMS[dev.index if dev.index is not None else 0] = \
            [torch.cuda.current_stream(dev)] + ([side] if side is not None else [])
    da, dw_head, db_head = head_backward((dest(net.predictor.weight), dest(net.predictor.bias)))
    done(net.predictor.weight, dw_head)
    done(net.predictor.bias, db_head)

    def bn_unchanged(rec):
        return rec["bn_ver"] == (rec["blk"].bn.weight._version, rec["blk"].bn.bias._version)

    def block_bwd(rec, da, need_dx=True, da_stats=None, producer=None):
        """da_stats: the two BatchNorm-backward sums of THIS block per pixel tile, taken in the epilogue of the data-gradient launch
        that produced `da` (then one pass over (dA, z) is left).  producer: the record of the block whose activation is this block's
        input -- when given (and the plain Winograd data gradient runs), this block's dX launch takes ITS sums the same way."""
        blk = rec["blk"]
        # ReLU mask recomputed from z (bit-identical to a > 0): the passes read two activation tensors instead of three.
        # That needs the forward's gamma / beta; if either was modified in place since, the mask comes from a itself.
        same = bn_unchanged(rec)
        if da_stats is not None:
            dz, dgamma, dbeta = ops.bn_relu_backward_tiles(da, rec["z"], blk.bn.weight.detach(), blk.bn.bias.detach(), rec["mean"],
                                                           rec["invstd"], da_stats, out=(dest(blk.bn.weight), dest(blk.bn.bias)))
        else:
            dz, dgamma, dbeta = ops.bn_relu_backward(da, None if same else rec["a"], rec["z"], blk.bn.weight.detach(), rec["mean"],
                                                     rec["invstd"], beta=blk.bn.bias.detach(), out=(dest(blk.bn.weight), dest(blk.bn.bias)))
        done(blk.bn.weight, dgamma)
        done(blk.bn.bias, dbeta)

        def wgrad():
            out = dest(blk.conv.weight)                   # (the reducer's bucket view, or None: a fresh tensor)
            if rec["up"] and rec["x1"] is not None:       # decoder entry: upsampled channels at the low resolution
                return ops.conv3x3_wgrad_up2x(rec["x0"], rec["x1"], dz, out=out)
            if rec["x1"] is None and not rec["up"] and tuning.use_winograd_wgrad(
                    int(rec["x0"].shape[1]), blk.conv.out_dim, int(dz.shape[2]), int(dz.shape[3])):
                # plain layer: Winograd-form weight gradient (tuning.WGRAD_WINO_TAIL, default 0: the no-role kernel for the first
                # blocks, whose launches are the last of backward -- measured, no gain)
                tail = rec["idx"] < tuning.WGRAD_WINO_TAIL and tuning.WGRAD_WINO_VARIANT < 0
                return ops.conv3x3_wgrad_wino(rec["x0"], dz, variant=5 if tail else None, out=out)
            return ops.conv3x3_wgrad(rec["x0"], dz, src1=rec["x1"], up0=rec["up"], out=out)

        if side is None:
            dw = wgrad()
            done(blk.conv.weight, dw)
        else:
            ready = torch.cuda.Event()
            ready.record(main)                                   # dZ (and, first time round, the activations) are final
            with torch.cuda.stream(side):
                side.wait_event(ready)
                dw = wgrad()
                done(blk.conv.weight, dw)                        # the hook's bucket copy is ordered on the side stream
            keep.append(dz)      # read by the side stream: stays alive until the main stream has joined it (below), so the
                                 # allocator can never hand its memory to later main-stream work too early
        if not need_dx:
            return None, None, None
        c0 = int(rec["x0"].shape[1])
        c1 = int(rec["x1"].shape[1]) if rec["x1"] is not None else 0
        n, _, h, w = dz.shape
        if rec["up"] and c1:
            # decoder entry: the gradient of the upsampled operand straight at the low resolution (4x4 stride-2
            # correlation of dZ, 4/9 of the MACs, no full-resolution intermediate), the skip half as a plain 3x3 dgrad
            skip_wino = tuning.use_winograd(blk.conv.out_dim, c1, int(h), int(w))
            w_skip_t = None
            low_stats = None
            dv = ops.dgrad_up2x_wino_variant()
            if dv == 2 and not ops.dgrad_up2x_wino_supported(c0, blk.conv.out_dim, int(h) // 2, int(w) // 2, 2):
                dv = 0
            if tuning.UP2X_WINO and ops.dgrad_up2x_wino_supported(c0, blk.conv.out_dim, int(h) // 2, int(w) // 2, dv):
                # 25 of the 36 F(4x4) products (variant 2), or one GEMM with K = 9 * Cout
                if (dv == 2 and producer is not None and tuning.BN_BWD_STATS_IN_DGRAD_UP2X and bn_unchanged(producer)
                        and producer["a"] is rec["x0"]):
                    # round 6: d_low IS the producer block's dA -- the launch takes its BatchNorm-backward sums from the write-out
                    pb = producer["blk"].bn
                    d_low, low_stats = ops.dgrad_up2x_wino_bnstats(dz, blk.packed_dgrad_up2x_wino(c0, dv), c0, producer["z"], producer["mean"],
                                                                   producer["invstd"], pb.weight.detach(), pb.bias.detach())
                else:
                    d_low = ops.dgrad_up2x_wino(dz, blk.packed_dgrad_up2x_wino(c0, dv), c0, variant=dv)
                if not skip_wino:
                    w_skip_t = blk.packed_dgrad_up2x(c0)[1]
            else:
                g_low, w_skip_t = blk.packed_dgrad_up2x(c0)
                d_low = ops.dgrad_up2x(dz, g_low, c0)
            def skip_dgrad():
                if skip_wino and tuning.use_wino43_dgrad(blk.conv.out_dim, c1, int(h), int(w)):
                    return ops.conv3x3_wino43(dz, blk.packed_wino43_t(c0), c1)
                if skip_wino:
                    return ops.conv3x3_wino(dz, blk.packed_wino_t(c0), c1)
                cfg = tuning.conv_config(c1, blk.conv.out_dim, int(n), int(h), int(w))
                return ops.conv3x3_dgrad(dz, w_skip_t, c1, 0, cfg=cfg)[0]

            if side is not None and tuning.DSKIP_SIDE:
                # off the critical chain: the skip half's gradient is consumed by the max-pool backward of the matching down block, several
                # layers later -- it runs on the side stream (behind this entry's weight gradient; dZ is kept alive for it like for that)
                if skip_wino:                              # (the filter panel is built on the main stream, where the optimiser changed the weights)
                    (blk.packed_wino43_t if tuning.use_wino43_dgrad(blk.conv.out_dim, c1, int(h), int(w)) else blk.packed_wino_t)(c0)
                packed = torch.cuda.Event()
                packed.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(packed)
                    d_skip = skip_dgrad()
                    done_ev = torch.cuda.Event()
                    done_ev.record(side)
                side_results.append((d_skip, done_ev))
            else:
                d_skip = skip_dgrad()
            return d_low, d_skip, low_stats
        if c1 == 0 and not rec["up"] and tuning.use_winograd(blk.conv.out_dim, c0, int(h), int(w)):
            # plain layer: dX = conv3x3(dZ, W^T flipped) is itself a plain 3x3 convolution -> the Winograd kernel
            if (producer is not None and tuning.BN_BWD_STATS_IN_DGRAD and tuning.wino_has_stats() and bn_unchanged(producer)
                    and producer["z"].shape[1] == c0):
                pb = producer["blk"].bn
                c4 = ops.bn_bwd_consts(producer["mean"], producer["invstd"], pb.weight.detach(), pb.bias.detach())
                dx, st = ops.conv3x3_wino_dgrad_bnstats(dz, blk.packed_wino_t(), c0, producer["z"], c4)
                return dx, None, st
            if tuning.use_wino43_dgrad(blk.conv.out_dim, c0, int(h), int(w)):
                if (producer is not None and tuning.BN_BWD_STATS_IN_DGRAD43 and bn_unchanged(producer) and producer["z"].shape[1] == c0
                        and ops.wino43_variant(None) != 1):
                    # round 6: this launch's write-out also takes the PRODUCER block's two BatchNorm-backward sums (it reads that block's z beside
                    # the dA it writes): the separate sums pass over (dA, z) is gone, one pass (the apply) is left
                    pb = producer["blk"].bn
                    dx, st = ops.conv3x3_wino43_dgrad_bnstats(dz, blk.packed_wino43_t(), c0, producer["z"], producer["mean"], producer["invstd"],
                                                              pb.weight.detach(), pb.bias.detach())
                    return dx, None, st
                return ops.conv3x3_wino43(dz, blk.packed_wino43_t(), c0), None, None
            return ops.conv3x3_wino(dz, blk.packed_wino_t(), c0), None, None
        if c1 == 0 and c0 % 64:
            # gradient w.r.t. the network INPUT (9 / 27 channels; only when the caller asked for it -- train.py never does):
            # the data-gradient kernels produce channel blocks of 64, so run it on the filter zero-padded to 64 input
            # channels and keep the first c0 planes
            cpad = (c0 + 63) // 64 * 64
            wpad = torch.zeros((blk.conv.out_dim, cpad, 3, 3), dtype=torch.float32, device=dz.device)
            wpad[:, :c0] = blk.conv.weight.detach()
            dxp, _ = ops.conv3x3_dgrad(dz, ops.pack_conv3x3_weights(wpad, transpose_flip=True), cpad, 0)
            return dxp[:, :c0].contiguous(), None, None
        cfg = tuning.conv_config(c0 + c1, blk.conv.out_dim, int(n), int(h), int(w))
        return ops.conv3x3_dgrad(dz, blk.packed_weight(transpose_flip=True), c0, c1, cfg=cfg) + (None,)

    x1, x2, x3 = ctx.skips
    idx = len(saved) - 1

    def from_side(t):
        """A tensor the side stream produced (a skip half's data gradient): the main stream waits for its event before reading it; the tensor was
        allocated on the side stream's pool and is handed to main-stream work, so the allocator is told."""
        for k, (ts, ev) in enumerate(side_results):
            if ts is t:
                main.wait_event(ev)
                t.record_stream(main)
                del side_results[k]
                break
        return t

    carried = {"stats": None}        # the sums a chain's LAST data gradient took for the block in front of the chain (decoder entries)

    def chain_bwd(count, da, first_needs_dx=True, stats=None):
        nonlocal idx
        d_skip = None
        if stats is None:
            stats, carried["stats"] = carried["stats"], None
        for k in range(count):
            rec = saved[idx]
            idx -= 1
            last = (k == count - 1)
            # inside a Double / Triple block the next record is the producer of this block's input: its BatchNorm-backward sums
            # come out of this block's data-gradient epilogue
            # ... and the record in front of a decoder entry (the previous chain's last block) is the producer of the entry's upsampled operand
            entry = last and rec["up"] and rec["x1"] is not None and idx >= 0
            da, d_skip, stats = block_bwd(rec, da, need_dx=(first_needs_dx or not last), da_stats=stats,
                                          producer=(saved[idx] if (not last or entry) else None))
        carried["stats"] = stats
        return da, d_skip

    # up_block_3 (2) -> dUp(128ch, full res), dSkip(x1)
    da, d_x1 = chain_bwd(2, da)                        # (the first block of each chain returns the gradient of the
    da, d_x2 = chain_bwd(2, da)                        # up_block_2      low-resolution operand of nn.Upsample directly)
    da, d_x3 = chain_bwd(3, da)                        # up_block_1
    def pool_bwd(x, d_pool, d_skip):
        """Gradient of a down block's output x = ReLU(BN(z)) (pooled below, concatenated into the decoder): the routed pool gradient + the skip
        half's.  Round 6: the pass also takes the block's last layer's BatchNorm-backward sums (it reads that layer's z instead of x)."""
        rec = saved[idx]                                   # the down block's last layer
        n, _, h, w = (int(v) for v in x.shape)
        if (tuning.BN_BWD_STATS_IN_POOL and bn_unchanged(rec) and rec["a"] is x and ops.maxpool2x2_bnstats_supported(n, h, w)):
            bn = rec["blk"].bn
            return ops.maxpool2x2_backward_add_bnstats(rec["z"], d_pool, d_skip, rec["mean"], rec["invstd"], bn.weight.detach(), bn.bias.detach())
        return ops.maxpool2x2_backward_add(x, d_pool, d_skip), None

    d_pool3, _ = chain_bwd(3, da)                      # bottleneck -> gradient of pool(x3)
    da, st = pool_bwd(x3, d_pool3, from_side(d_x3))
    d_pool2, _ = chain_bwd(3, da, stats=st)            # down_block_3
    da, st = pool_bwd(x2, d_pool2, from_side(d_x2))
    d_pool1, _ = chain_bwd(2, da, stats=st)            # down_block_2
    da, st = pool_bwd(x1, d_pool1, from_side(d_x1))
    dx, _ = chain_bwd(2, da, first_needs_dx=ctx.need_dx, stats=st)   # down_block_1
    if side is not None:
        main.wait_stream(side)                                   # every weight gradient is final for whoever comes next
    keep.clear()
    ctx.saved = ctx.skips = ctx.head_in = None
    if _backward_end_hook is not None:
        _backward_end_hook()
    return (dx if ctx.need_dx else None), tuple(grads.get(id(p)) for p in net._train_params)


class _TrackNetTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        # params are listed only so that autograd routes their gradients; values are read from the module.
        saved, y, skips = _train_forward_body(net, x)
        p = ops.head1x1_sigmoid(y, net.predictor.weight.detach(), net.predictor.bias.detach())
        # The OUTPUT goes through save_for_backward: an output kept as a plain ctx attribute is a reference cycle
        # (p.grad_fn -> ctx -> p) that pins every activation of the step until the cyclic GC runs (measured: +0.85 GB/step).
        if hasattr(ctx, "save_for_backward"):
            ctx.save_for_backward(p)
        ctx.net, ctx.saved, ctx.head_in = net, saved, y
        ctx.skips = skips
        ctx.need_dx = x.requires_grad
        return p

    @staticmethod
    def backward(ctx, dp):
        dp = dp.contiguous()
        (p_out,) = ctx.saved_tensors
        net, head_in = ctx.net, ctx.head_in
        dx, pgrads = _train_backward_body(ctx, dp.device, lambda out: ops.head_backward(dp, p_out, head_in, net.predictor.weight.detach(), out=out))
        return (None, dx) + pgrads


class _TrackNetTrainLoss(torch.autograd.Function):
    """forward(train) + WBCELoss as ONE node with sigmoid + WBCE fused into the head in both directions (the north-star's
    "sigmoid+WBCE-loss fused into the heatmap head"): the forward writes p once and takes the loss from the registers that hold
    it; the backward forms dL/dp inside the head's backward from (p, y) -- no dP tensor exists.  Returns (loss, p); p is
    marked non-differentiable (it is an observation, the loss is what is differentiated)."""

    @staticmethod
    def forward(ctx, net, x, y, reduce, *params):
        saved, head_in, skips = _train_forward_body(net, x)
        p, loss = ops.head1x1_sigmoid_wbce(head_in, net.predictor.weight.detach(), net.predictor.bias.detach(), y, reduce)
        ctx.save_for_backward(p, y)
        ctx.mark_non_differentiable(p)
        ctx.net, ctx.saved, ctx.head_in, ctx.skips = net, saved, head_in, skips
        ctx.need_dx, ctx.reduce = x.requires_grad, bool(reduce)
        return (loss.reshape(()) if reduce else loss), p

    @staticmethod
    def backward(ctx, dloss, _dp_unused):
        p_out, y = ctx.saved_tensors
        net, head_in = ctx.net, ctx.head_in
        up = dloss.reshape(-1).contiguous().float()
        dx, pgrads = _train_backward_body(
            ctx, up.device, lambda out: ops.head_wbce_backward(y, p_out, head_in, net.predictor.weight.detach(), up, ctx.reduce, out=out))
        return (None, dx, None, None) + pgrads


def tracknet_forward_loss(net, x, y, reduce=True):
    """(loss, y_pred) of one training forward: `y_pred = net(x); loss = WBCELoss(y_pred, y, reduce)` (train.py:92-93) with the
    loss fused into the head.  Differentiable through `loss` (parameters and, if requested, x)."""
    if not net.training:
        raise RuntimeError("tracknet_forward_loss is the training-mode path: call net.train() first")
    x = x.contiguous()
    y = y.to(torch.float32).contiguous()
    params = [p for p in net.parameters()]
    net._train_params = params
    return _TrackNetTrainLoss.apply(net, x, y, bool(reduce), *params)


def tracknet_forward_train(net, x):
    params = [p for p in net.parameters()]
    net._train_params = params
    if not torch.is_grad_enabled() or not (x.requires_grad or any(p.requires_grad for p in params)):
        # train-mode forward without autograd (e.g. under no_grad): still uses batch statistics / updates buffers
        class _Ctx:
            pass
        return _TrackNetTrain.forward(_Ctx(), net, x, *params)
    return _TrackNetTrain.apply(net, x, *params)


class _WBCELoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y, reduce):
        y_pred, y = y_pred.contiguous(), y.contiguous()
        out = ops.wbce_forward(y_pred, y, reduce)
        ctx.save_for_backward(y_pred, y)
        ctx.reduce = reduce
        return out.reshape(()) if reduce else out

    @staticmethod
    def backward(ctx, g):
        y_pred, y = ctx.saved_tensors
        return ops.wbce_backward(y_pred, y, g.reshape(-1).contiguous().float(), ctx.reduce), None, None


def wbce_loss(y_pred, y, reduce=True):
    return _WBCELoss.apply(y_pred, y.to(y_pred.dtype), bool(reduce))


class _InpaintNetTrain(torch.autograd.Function):
    """InpaintNet.forward (model.py:113-129) + its backward as one autograd node over the conv1d HIP kernels."""

    @staticmethod
    def forward(ctx, net, x, m, *params):
        p = [(w.detach(), b.detach()) for w, b in net.conv_params()]
        x1 = ops.conv1d_k3(x, *p[0], src1=m, src_nlc=True)
        x2 = ops.conv1d_k3(x1, *p[1])
        x3 = ops.conv1d_k3(x2, *p[2])
        b1 = ops.conv1d_k3(x3, *p[3])
        b2 = ops.conv1d_k3(b1, *p[4])
        u1 = ops.conv1d_k3(b2, *p[5], src1=x3)
        u2 = ops.conv1d_k3(u1, *p[6], src1=x2)
        u3 = ops.conv1d_k3(u2, *p[7], src1=x1)
        out = ops.conv1d_k3(u3, *p[8], dst_nlc=True, act=ops.ACT_SIGMOID)
        if hasattr(ctx, "save_for_backward"):
            ctx.save_for_backward(out)                   # the output: never as a plain attribute (reference cycle)
        ctx.net, ctx.acts, ctx.inp = net, (x1, x2, x3, b1, b2, u1, u2, u3), (x, m)
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        x1, x2, x3, b1, b2, u1, u2, u3 = ctx.acts
        (out,) = ctx.saved_tensors
        x, m = ctx.inp
        w = [wb[0].detach() for wb in net.conv_params()]
        g = {}

        def layer(i, dpre, src0, src1=None, src_nlc=False):
            dw, db = ops.conv1d_k3_wgrad(src0, dpre, src1=src1, src_nlc=src_nlc)
            g[i] = (dw, db)

        L = ops.ACT_LEAKY_RELU
        d = ops.conv1d_act_backward(dout.contiguous(), out, ops.ACT_SIGMOID, nlc=True)        # predictor
        layer(8, d, u3)
        d_u3, _ = ops.conv1d_k3_dgrad(d, w[8], 32)
        d = ops.conv1d_act_backward(d_u3, u3, L)                                               # up_3: cat([u2, x1])
        layer(7, d, u2, x1)
        d_u2, d_x1 = ops.conv1d_k3_dgrad(d, w[7], 64, 32)
        d = ops.conv1d_act_backward(d_u2, u2, L)                                               # up_2: cat([u1, x2])
        layer(6, d, u1, x2)
        d_u1, d_x2 = ops.conv1d_k3_dgrad(d, w[6], 128, 64)
        d = ops.conv1d_act_backward(d_u1, u1, L)                                               # up_1: cat([b2, x3])
        layer(5, d, b2, x3)
        d_b2, d_x3 = ops.conv1d_k3_dgrad(d, w[5], 256, 128)
        d = ops.conv1d_act_backward(d_b2, b2, L)                                               # buttleneck.conv_2
        layer(4, d, b1)
        d_b1, _ = ops.conv1d_k3_dgrad(d, w[4], 256)
        d = ops.conv1d_act_backward(d_b1, b1, L)                                               # buttleneck.conv_1
        layer(3, d, x3)
        ops.conv1d_k3_dgrad(d, w[3], 128, dx0=d_x3)                                            # += skip gradient of x3
        d = ops.conv1d_act_backward(d_x3, x3, L)                                               # down_3
        layer(2, d, x2)
        ops.conv1d_k3_dgrad(d, w[2], 64, dx0=d_x2)
        d = ops.conv1d_act_backward(d_x2, x2, L)                                               # down_2
        layer(1, d, x1)
        ops.conv1d_k3_dgrad(d, w[1], 32, dx0=d_x1)
        d = ops.conv1d_act_backward(d_x1, x1, L)                                               # down_1 (inputs need no grad)
        layer(0, d, x, m, src_nlc=True)
        out_g = [None, None, None]
        for i in range(9):
            out_g.extend(g[i])
        return tuple(out_g)


class _InpaintNetFusedTrain(torch.autograd.Function):
    """The same node as three launches: fused forward that saves the hidden activations, one fused data-gradient kernel (a sequence's
    gradient stays in LDS through all nine layers) and one launch for every dW / db (csrc/kernels/inpaint_fused_train.h)."""

    @staticmethod
    def forward(ctx, net, x, m, *params):
        from . import inpaint_ops
        packed = inpaint_ops.packed_params(net)
        out, acts = ops.inpaintnet_fused_train_forward(x, m, packed)
        if hasattr(ctx, "save_for_backward"):
            ctx.save_for_backward(out)
        ctx.net, ctx.acts, ctx.inp, ctx.packed = net, acts, (x, m), packed
        ctx.shapes = [tuple(p.shape) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import inpaint_ops
        (out,) = ctx.saved_tensors
        x, m = ctx.inp
        flat = ops.inpaintnet_fused_backward(x, m, dout.contiguous().float(), out, ctx.acts, ctx.packed, inpaint_ops.packed_params_t(ctx.net))
        grads, off = [None, None, None], 0
        for shp in ctx.shapes:                                  # views into the flat buffer, in state_dict order (weight, bias per layer)
            n = 1
            for v in shp:
                n *= v
            grads.append(flat[off:off + n].view(shp))
            off += n
        ctx.acts = ctx.inp = ctx.packed = None
        return tuple(grads)


def inpaintnet_forward_train(net, x, m):
    from . import inpaint_ops
    x = x.contiguous().float()
    m = m.contiguous().to(torch.float32)
    flat = []
    for wt, b in net.conv_params():
        flat += [wt, b]
    if inpaint_ops.use_fused_train(int(x.shape[0]), int(x.shape[1])):
        return _InpaintNetFusedTrain.apply(net, x, m, *flat)
    return _InpaintNetTrain.apply(net, x, m, *flat)
