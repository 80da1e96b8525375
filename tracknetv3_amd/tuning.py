"""Per-layer tile-configuration table for the MFMA conv kernel.

``conv_tuning.json`` is produced on an MI355X by ``python bench.py --tune`` (it times every compiled tile
configuration on every distinct conv shape of the workload and keeps the fastest).  Without an entry the library's
own heuristic (cfg = -1) is used.  Keys: "cout,cin,n,h,w".
"""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tuning.json")
_table = None


def _load():
    global _table
    if _table is None:
        _table = {}
        if os.path.exists(_PATH):
            with open(_PATH) as f:
                _table = {k: int(v) for k, v in json.load(f).get("configs", {}).items()}
    return _table


def conv_config(cout, cin, n, h, w):
    return _load().get(f"{cout},{cin},{n},{h},{w}", -1)


def save(configs, meta=None):
    global _table
    with open(_PATH, "w") as f:
        json.dump({"meta": meta or {}, "configs": configs}, f, indent=1, sort_keys=True)
    _table = None


def _choice(name, default, allowed, what):
    """An integer kernel-variant knob from the environment, checked HERE against what the product library dispatches (ABI 6).  A stale value
    (a twin that left the product library) used to surface as 'unknown kernel variant' from inside the first launch; raising at import
    instead made the WHOLE package unimportable for it -- inference paths that never touch the knob included.  Now: a warning,
    and the default."""
    raw = os.environ.get(name)
    if raw is None:
        return default
    try:
        v = int(raw)
    except ValueError:
        v = None
    if v not in allowed:
        import warnings
        warnings.warn(f"{name}={raw!r} ignored: {what} dispatches {sorted(allowed)} (the measured-and-rejected generations live in the diagnostics "
                      f"library; the scripts/ loader) -- using {default}", RuntimeWarning, stacklevel=2)
        return default
    return v


# Eval-mode plain layers run in Winograd F(2x2, 3x3) form when the shape qualifies (ops.wino_supported) and the
# contraction is deep enough for the transforms to pay: measured on MI355X at batch 10 (scripts/wino_sweep.py) it wins
# 1.3-1.5x on every 64..512-channel layer and 1.2x on the 27-channel stem; thinner inputs stay on the direct kernel.
WINOGRAD = os.environ.get("TNV3_WINOGRAD", "1") != "0"      # TNV3_WINOGRAD=0: direct 3x3 kernels everywhere (A/B knob)
WINOGRAD_MIN_CIN = int(os.environ.get("TNV3_WINO_MIN_CIN", "24"))      # measured: the 27-channel stem gains too (0.52 -> 0.42 ms)


WINOGRAD_MIN_SKIP = int(os.environ.get("TNV3_WINO_MIN_SKIP", "64"))     # skip half of a decoder entry (addend read in the epilogue)


# Eval forward: plain layers (and the skip halves of the decoder entries) in Winograd F(4x4, 3x3) form where the shape allows it
# (Cout % 64 == 0, H % 4 == 0, W % 64 == 0: every level of the 288x512 network) -- 36 products per 4x4 output tile instead of
# F(2x2)'s 16 per 2x2: 1.2x (the 27-channel stem) to 1.6x (256 channels) faster per layer (profiles/r03_wino43_ab.json), at 1.0-1.6e-6 of the output scale per layer instead of 3-6e-7 (the whole network's heat maps stay
# within 1.7e-6 of the fp64 forward, the direct fp32 forward's level: profiles/r03_wino_f43_precision.json).  TNV3_WINO43=0: F(2x2) everywhere.
WINO43 = os.environ.get("TNV3_WINO43", "1") != "0"
WINO43_MIN_CIN = int(os.environ.get("TNV3_WINO43_MIN_CIN", "16"))
# Which geometry of the F(4x4) kernel (kernels/conv3x3_wino43s_mfma.h: 16x16x4 MFMAs, all 36 transform coefficients of a block in one wave, the
# output transform in registers): 0 = 128 output channels x one tile row where Cout % 128 == 0, else 64 x two tile rows; 2 = the 64-channel
# geometry always.  (1, the 32x32x2 predecessor kernels/conv3x3_wino43_mfma.h, is a twin of the diagnostics library since ABI 6.)
WINO43_VARIANT = _choice("TNV3_WINO43_VARIANT", 0, {0, 2}, "tnv3_conv3x3_wino43_forward")
# MaxPool2d(2, 2) behind the down blocks' last layers as a second output of the F(4x4) kernel's write-out (variants 0 / 2; bit-identical to the
# separate pass).  TNV3_FUSE_POOL=0: the separate maxpool2x2 launches.
FUSE_POOL = os.environ.get("TNV3_FUSE_POOL", "1") != "0"


# Training: the plain layers' data gradients (and the skip halves') through the same F(4x4, 3x3) kernel (the data gradient IS a plain
# 3x3 convolution of dZ with the transposed, flipped filter).  TNV3_WINO43_DGRAD=0: the F(2x2) kernels.
WINO43_DGRAD = os.environ.get("TNV3_WINO43_DGRAD", "1") != "0"


def use_wino43_dgrad(cin, cout, h, w):
    return WINO43_DGRAD and use_wino43(cin, cout, h, w)


# The TRAINING forward through the F(4x4) kernel (its epilogue takes BatchNorm's batch statistics like the F(2x2) kernels').  Batch-statistics
# BatchNorm amplifies the forward's per-layer rounding, so this is the configuration with the least margin: at 288x512 the training-mode
# heat maps sit 4-5e-5 from the fp64 oracle (bar 1e-4; F(2x2) forward: 1.6e-5) and the gradients at ~1.6x torch-fp32's own distance from
# fp64 (profiles/r04_fullsize_train_parity_*.json: batch 2 with either forward, batch 10 with this one).  Default since round 4: the
# driver's GPU suite runs every training parity test with both forwards (tests/conftest.py::train_fwd).  TNV3_WINO43_TRAIN=0: F(2x2).
WINO43_TRAIN = os.environ.get("TNV3_WINO43_TRAIN", "1") != "0"


# ... and at WHICH resolution levels (rows of the layer's output): round 5 measured what the F(4x4) forward at one level only adds to the
# training-mode heat-map error (1.4 / 1.1 / 1.0 / 0.4e-5 at 288 / 144 / 72 / 36 rows over F(2x2)'s 1.6e-5, gain 2.4); round 6 drew the Pareto of
# error against milliseconds over all 16 subsets (tests/studies/train_precision_pareto.py, profiles/r06_train_precision_pareto.json) and the
# default is the cheapest subset with a 2x margin under the 1e-4 bar at a trained-like logit range (<= 3.5e-5 at head gain 2.4, <= 5e-5 at 4).
# TNV3_WINO43_TRAIN_LEVELS="288,144,72,36": everywhere (round 4 / 5's default); "": nowhere (= TNV3_WINO43_TRAIN=0).  Levels of other image
# sizes (a 64x128 test network: 64 / 32 / 16 / 8 rows) are not in the set and take F(4x4) -- the set names where it is NOT trusted at full size.
_ALL_LEVELS_288 = (288, 144, 72, 36)
WINO43_TRAIN_LEVELS = frozenset(int(v) for v in os.environ.get("TNV3_WINO43_TRAIN_LEVELS", "288,144,72,36").split(",") if v.strip())


# ... or per LAYER (forward order 0 .. 16: down_block_1.conv_1 = 0, ..., bottleneck = 7-9, ..., up_block_3.conv_2 = 16): the layers listed here run
# their training forward in F(2x2) form whatever their level.  The error of the heat maps does not come from the layers evenly -- switching the
# stem or down_block_1.conv_2 alone removes 0.8-1.1e-5 of the 4.5e-5, no other single layer more than the seed-to-seed noise
# (tests/studies/train_precision_layers.py, profiles/r06_train_precision_layers.json) -- so two layers buy what a whole level costs.  TNV3_WINO43_TRAIN_F22_LAYERS="": none.
# Default since round 6: the first block (layers 0, 1 -- the un-normalised image planes and the first activation: positive data with a large
# mean, where F(4x4)'s transforms cancel most) in F(2x2), everything else in F(4x4), the decoder entries' upsampled halves in the 25-of-36 form
# (UP2X_WINO_VARIANT_TRAIN = 2): worst of 3 seeds 4.2e-5 / 5.7e-5 / 8.3e-5 at head gain 2.4 / 4 / 6 at +0.1 ms per step over round 5's default,
# which measured 4.5e-5 / 6.7e-5 / 1.1e-4 (profiles/r06_train_precision_sets.json).  More margin costs time roughly linearly:
# TNV3_WINO43_TRAIN_LEVELS=72,36 -> 2.7e-5 / 4.1e-5 / 5.9e-5 for +2.0 ms; TNV3_WINO43_TRAIN=0 (F(2x2) everywhere) -> 1.6e-5 / 2.8e-5 / 3.7e-5
# (torch-fp32's own level) for +3.5 ms (profiles/r06_train_precision_pareto.json).
WINO43_TRAIN_F22_LAYERS = frozenset(int(v) for v in os.environ.get("TNV3_WINO43_TRAIN_F22_LAYERS", "0,1").split(",") if v.strip())


def use_wino43_train(cin, cout, h, w, layer=None):
    if int(h) in _ALL_LEVELS_288 and int(h) not in WINO43_TRAIN_LEVELS:
        return False
    if layer is not None and int(layer) in WINO43_TRAIN_F22_LAYERS:
        return False
    return WINO43_TRAIN and BN_STATS_IN_EPILOGUE and use_wino43(cin, cout, h, w)


def use_wino43(cin, cout, h, w):
    if not (WINOGRAD and WINO43) or cin < WINO43_MIN_CIN:
        return False
    from . import ops
    return ops.wino43_supported(cin, cout, h, w)


def use_winograd(cin, cout, h, w):
    if not WINOGRAD or cin < WINOGRAD_MIN_CIN:
        return False
    from . import ops
    return ops.wino_supported(cin, cout, h, w)


# The Winograd-form weight gradient is 1.35-1.5x faster than the direct kernel on every plain layer shape from 64
# channels up (scripts/wgrad_wino_sweep.py).  (It used to lose on the 64-channel layers -- because of its split-K fold,
# not the MFMA kernel: a few blocks walking hundreds of slabs serially; fixed in wgrad_wino_fold_kernel.)
WINOGRAD_WGRAD_MIN_CH = int(os.environ.get("TNV3_WINO_WGRAD_MIN_CH", "64"))       # output channels
WINOGRAD_WGRAD_MIN_CIN = int(os.environ.get("TNV3_WINO_WGRAD_MIN_CIN", "1"))     # input channels when not a multiple of 64 (65 = never)


def use_winograd_wgrad(cin, cout, h, w):
    """Kernels 5 / 8 (what -1 picks when Cin % 64 != 0) take any Cin -- a partial block of 64 input channels: the stem layer (Cin = 27)
    costs a 64-channel layer's 0.55 ms instead of the direct kernel's 0.96 ms, 0.36 ms of the step --; the older generations need
    64-multiples on both sides."""
    if not WINOGRAD or cout < WINOGRAD_WGRAD_MIN_CH:
        return False
    if cin % 64 and (WGRAD_WINO_VARIANT not in (-1, 5, 8) or cin < WINOGRAD_WGRAD_MIN_CIN):
        return False
    from . import ops
    return ops.wgrad_wino_supported(cin, cout, h, w)


def use_wino43_wgrad(cin, cout, h, w):
    """Whether the Winograd-form weight gradient of this layer runs the F(4x4) kernel (variant 8: the library's pick for -1 where H % 4 == 0)."""
    return use_winograd_wgrad(cin, cout, h, w) and WGRAD_WINO_VARIANT in (-1, 8) and h % 4 == 0


# The weight gradients of the first WGRAD_WINO_TAIL Conv2DBlocks (forward order: stem, down_block_1.conv_2, down_block_2.conv_1, ...)
# are the last launches of backward, when the main stream is winding down -- the case where the no-role kernel (variant 5: +12 % per
# call, full-CU footprint) might win.  Measured (scripts/train_tail_ab.sh, two repeats): 0 blocks 31.66 / 31.81 ms per step, 2: 31.74 /
# 31.92, 4: 31.77 / 31.90, 7: 32.05 / 31.96, 10: 31.96 / 31.89 -- it does not; default 0.
WGRAD_WINO_TAIL = int(os.environ.get("TNV3_WGRAD_WINO_TAIL", "0"))


# HIP priority of the side stream the weight gradients run on (0 = default, -1 = high: its workgroups are dispatched before the main
# stream's when CUs free up).  Measured (scripts/train_prio_ab.sh): 31.59 / 31.73 ms per step at 0, 31.75 / 31.59 at -1 -- no effect:
# both streams are saturated (profiles/r03_train_timeline.json), priority only reorders who waits.
WGRAD_STREAM_PRIORITY = int(os.environ.get("TNV3_WGRAD_STREAM_PRIORITY", "0"))
# The data gradient of a decoder entry's SKIP half (d_x3 / d_x2 / d_x1) is not on backward's critical chain: its consumer is the max-pool
# backward of the matching down block, three to nine layers later.  With the round-6 weight-gradient kernels the side stream has slack, so
# these three launches (~0.9 ms of MFMA time at 288x512, batch 10) CAN run there, behind the entry's weight gradient (TNV3_DSKIP_SIDE=1).
# Measured and rejected as a default (profiles/r06_train_step_ab.txt, two repeats in one session): 22.04 / 22.60 ms with it, 21.81 / 21.91
# without -- the side stream's MFMA kernels take CUs from the main chain's, and the chain does not get shorter by what moved.
DSKIP_SIDE = os.environ.get("TNV3_DSKIP_SIDE", "0") == "1"


# Kernel-family choices are per-call arguments of the C ABI (no process-wide state inside the library); these are the
# defaults the Python layer passes.  -1 = the library's default, resolved INSIDE the library (kWinoDefaultVariant: today 5, the
# streaming persistent Winograd kernel; register-staged weight gradient) -- layout and capabilities of "-1" are queried from it
# (tnv3_conv3x3_wino_layout / tnv3_conv3x3_wino_has_stats), never assumed here.
# F(2x2) forward / data-gradient kernel (what runs where F(4x4) does not apply): -1 = the library's pick -- 6, the 128-channel form, where
# Cout % 128 == 0, else 5, the streaming persistent kernel.
WINO_VARIANT = _choice("TNV3_WINO_VARIANT", -1, {-1, 5, 6}, "tnv3_conv3x3_wino_forward")
WGRAD_VARIANT = _choice("TNV3_WGRAD_VARIANT", 0, {0}, "tnv3_conv3x3_wgrad")      # the register-staged direct kernel (the LDS-DMA twin: the diagnostics library)
# Winograd-form weight gradient of the plain layers (and the skip half of the decoder-entry layers): -1 = the library's pick -- the
# F(4x4) kernel (8) where H % 4 == 0, else F(2x2) kernel 1 (5 for the stem); 8 the F(4x4) kernel (other heights fall back to -1);
# 1 the role-split F(2x2) kernel, 5 the one where every wave streams and transforms (any Cin).
# Measured per call at batch 10 (profiles/r04_wgrad_wino43_ab.json): 8 is 1.45-1.51x faster than 1 on every plain shape (0.396 vs
# 0.598 ms at 64 -> 64 @ 288x512), 2.56x on the stem (0.237 vs 0.605 ms: blocks of 32 input channels instead of 64); the training step
# 26.11 -> 23.59 ms.  Its gradient is 2-4e-6 (max) / 3-6e-7 (rms) of max|dW| from the F(2x2) one.
WGRAD_WINO_VARIANT = _choice("TNV3_WGRAD_WINO_VARIANT", -1, {-1, 1, 5, 8}, "tnv3_conv3x3_wgrad_wino")


# Data gradient of the decoder entries' upsampled halves: 2 = the 25-of-36 F(4x4) form on the 16x16x4 kernel (kernels/conv3x3_wino43s_mfma.h
# MODE 2, round 5: 6.25 multiply-adds per low-resolution pixel; c0 % 64 == 0, else the next), 0 = the one-GEMM F(2x2) kernel (9; c0 % 128 == 0).
DGRAD_UP2X_WINO_VARIANT = _choice("TNV3_DGRAD_UP2X_WINO_VARIANT", 2, {-1, 0, 1, 2}, "tnv3_dgrad_up2x_wino")


# Weight gradient of the decoder entries' upsampled halves: -1 = the fastest form the shape allows -- 2, the 25-of-36 F(4x4) form
# (kernels/wgrad_up2x_wino43_mfma.h, round 5: 6.25 multiply-adds per low-resolution pixel, any c0), else 1, the 9-GEMM F(2x2) form
# (9; c0 % 128 == 0), else 0, four 2x2-window launches (16).  The weight gradient is a leaf: nothing amplifies its rounding.
WGRAD_UP2X_VARIANT = _choice("TNV3_WGRAD_UP2X_VARIANT", -1, {-1, 0, 1, 2}, "tnv3_conv3x3_wgrad_up2x (up_variant)")


# Training: all Winograd filter panels that the optimiser step made stale are rebuilt by one launch at the start of the forward
# (model.TrackNet.repack_wino_panels) instead of one 14-us launch in front of every convolution / data gradient.
WINO_REPACK_MULTI = os.environ.get("TNV3_WINO_REPACK_MULTI", "1") != "0"


# BatchNorm batch statistics from the convolution's epilogue (training forward): available in Winograd kernel variants 3, 4 and 5.
BN_STATS_IN_EPILOGUE = os.environ.get("TNV3_BN_STATS_EPILOGUE", "1") != "0"


# Inference: one batch is split over two HIP streams (6 : 4) -- its images are independent, and the second stream's launches fill
# the CUs that the tail of every per-layer launch leaves idle (720 tiles on 256 CUs = 2.8 rounds: the last one is 81 % full).
# Measured on the batch-10 288x512 forward: 9.91 -> 9.33 ms (profiles/r02_split_stream_probe.json); outputs are bit-identical.
# Other shares and three or four streams are all slower (profiles/r03_infer_split_sweep.txt: 6,4 8.07 ms; 7,3 8.09; 4,4,2 8.15; 5,5 8.22;
# re-swept on round 4's kernels: 6,4 5.27-5.30 ms; 7,3 5.30; 5,5 5.38; 4,3,3 5.46; 4,4,2 5.47; 5,3,2 5.48; 8,2 5.63; 3,3,2,2 5.72).
INFER_SPLIT = os.environ.get("TNV3_INFER_SPLIT", "1") != "0"
def _split_parts(text):
    """Shares of the batch, one stream each: positive integers, at least one -- anything else falls back to 6 : 4."""
    try:
        parts = tuple(int(v) for v in text.split(",") if v.strip())
    except ValueError:
        return (6, 4)
    return parts if parts and all(p > 0 for p in parts) else (6, 4)


INFER_SPLIT_PARTS = _split_parts(os.environ.get("TNV3_INFER_SPLIT_PARTS", "6,4"))
INFER_SPLIT_MIN_BATCH = 4
INFER_SPLIT_MIN_PIXELS = 1 << 19          # batch x H x W below which the launches are too short to be worth a second stream


# Upsampled half of the decoder-entry layers: Winograd form with 9 of the 16 GEMMs (kernels/conv_up2x_wino_mfma.h) instead of the
# four pre-summed 2x2 class filters (conv_up2x_mfma.h): 0.79 -> 0.50 ms per layer at batch 10 (profiles/r02_up2x_wino_ab.json).
UP2X_WINO = os.environ.get("TNV3_UP2X_WINO", "1") != "0"
# ... and which Winograd form: 0 = 9 of the 16 F(2x2) GEMMs (kernels/conv_up2x_wino_mfma.h), 2 = 25 of the 36 F(4x4) products on the 16x16x4
# kernel (kernels/conv3x3_wino43s_mfma.h MODE 1: Lavin's points, 1-5e-6 of the output scale from fp64 -- an addend of the skip half's launch).
# The eval forward and the training forward choose separately (batch-statistics BatchNorm amplifies the forward's rounding).
UP2X_WINO_VARIANT = _choice("TNV3_UP2X_WINO_VARIANT", 2, {-1, 0, 1, 2}, "tnv3_conv_up2x_wino_forward")
UP2X_WINO_VARIANT_TRAIN = _choice("TNV3_UP2X_WINO_VARIANT_TRAIN", 2, {-1, 0, 1, 2}, "tnv3_conv_up2x_wino_forward")


# BatchNorm + ReLU backward: the two per-channel sums of block L (sum g, sum g * xhat) taken in the epilogue of the Winograd data-gradient
# launch of block L + 1 -- which produces exactly dA_L -- instead of a pass over (dA, z): ten of the seventeen blocks (those followed by a
# plain conv inside their Double / Triple block).  Built, parity-green (tests/test_gpu_training.py::test_bn_backward_sums_from_the_data_
# gradient_epilogue) and MEASURED SLOWER: A/B in one session 32.30 / 32.53 ms per step without, 32.88 / 33.06 with
# (profiles/r03_bn_bwd_epilogue_ab.json) -- the 16 extra 8-byte z loads, the per-channel constants and the fp64 products per tile cost the
# MFMA kernel's epilogue (which nothing overlaps: one workgroup fills the CU) more than the HBM-bound pass they replace, which runs
# beside the other stream's kernels.  Default OFF; TNV3_BN_BWD_STATS_IN_DGRAD=1 switches it on.
BN_BWD_STATS_IN_DGRAD = os.environ.get("TNV3_BN_BWD_STATS_IN_DGRAD", "0") == "1"
# The same fusion on the F(4x4) 16x16x4 kernel (round 6; conv3x3_wino43s_kernel<.., STATS = 2>, tnv3_conv3x3_wino43_dgrad_bnstats): there the
# write-out is per-lane register arithmetic (no LDS exchange), the z rows are read like an addend's, and since the round-6 weight gradients the
# MAIN stream's chain (BatchNorm backward -> data gradient -> ...) is what bounds the step -- the sums pass sits on it, the epilogue's extra
# vector work costs 1-3 % of a data-gradient launch from 128 channels up (13 % on the two 64-channel launches).  Applies to the 10 of the 17 layers
# whose input is the previous layer's activation (inside a Double / Triple block).  TNV3_BN_BWD_STATS_IN_DGRAD43=0: the separate sums pass.
BN_BWD_STATS_IN_DGRAD43 = os.environ.get("TNV3_BN_BWD_STATS_IN_DGRAD43", "1") != "0"
# Round 6: ... and for the last layer of each down block (its gradient comes out of the max-pool backward + skip add, not out of a data
# gradient) from that pass (ops.maxpool2x2_backward_add_bnstats: it reads z instead of a, the sums cost no traffic).
# TNV3_BN_BWD_STATS_IN_POOL=0: the separate sums pass.
BN_BWD_STATS_IN_POOL = os.environ.get("TNV3_BN_BWD_STATS_IN_POOL", "1") != "0"
# ... and for the block in front of a decoder entry (the bottleneck's and the first two up blocks' last layers) from the upsampled half's data
# gradient (kernel variant 2 only: ops.dgrad_up2x_wino_bnstats) -- built, measured, OFF: these three blocks sit at the low resolutions, their
# separate sums passes are the cheap ones (~0.1 ms together), and the epilogue costs the 25-of-36 kernel more than that: 21.99 / 21.90 ms per
# step with it against 21.84 / 21.75 without (profiles/r06_train_step_ab.txt).  TNV3_BN_BWD_STATS_IN_DGRAD_UP2X=1 turns it on.
BN_BWD_STATS_IN_DGRAD_UP2X = os.environ.get("TNV3_BN_BWD_STATS_IN_DGRAD_UP2X", "0") == "1"
# Round 6: the forward twin -- the BatchNorm + ReLU pass of a down block's last layer also writes the pooled tensor (no separate pooling pass,
# no second read of a).  TNV3_POOL_IN_BN_APPLY=0: ops.maxpool2x2 behind it.
POOL_IN_BN_APPLY = os.environ.get("TNV3_POOL_IN_BN_APPLY", "1") != "0"


def wino_has_stats():
    from . import ops
    return BN_STATS_IN_EPILOGUE and ops.wino_variant_has_stats(WINO_VARIANT)
