"""Tensor-level wrappers over the C ABI (include/tracknetv3_hip.h).  Plumbing only: argument checks, output
allocation from the PyTorch caching allocator, current-stream hand-off.  All arithmetic is in the HIP library."""
import ctypes

import torch

from . import _lib

BN_EPS = 1e-5


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _lib.Tnv3Error(f"expected float32 tensor, got {t.dtype}")


def conv3x3_num_configs():
    return _lib.load().tnv3_conv3x3_num_configs()


def conv3x3_config_info(cfg):
    import ctypes
    v = [ctypes.c_int() for _ in range(6)]
    _lib.check(_lib.load().tnv3_conv3x3_config_info(cfg, *[ctypes.byref(x) for x in v]))
    return dict(zip(("m_block", "tile_rows", "tile_cols", "chan_chunk", "threads", "lds_bytes"), [x.value for x in v]))


def pack_conv3x3_weights(w, transpose_flip=False, out=None):
    """W[Cout][Cin][3][3] -> packed filter for tnv3_conv3x3_forward (forward, or data-gradient when transpose_flip)."""
    lib = _lib.load()
    _f32(w)
    _lib.dev_check(w, out)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3):
        raise _lib.Tnv3Error("pack_conv3x3_weights: expected a (Cout, Cin, 3, 3) filter")
    n = lib.tnv3_conv3x3_packed_floats(cout, cin, int(transpose_flip))
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    elif out.numel() != n:
        raise _lib.Tnv3Error("pack_conv3x3_weights: wrong output size")
    _lib.check(lib.tnv3_pack_conv3x3_weights(_lib.ptr(w), _lib.ptr(out), cout, cin, int(transpose_flip), _lib.stream_ptr(w)))
    return out


def bn_eval_scale(gamma, running_var, eps=BN_EPS):
    """scale = gamma / sqrt(running_var + eps) (eval-mode BatchNorm2d; mean and beta are passed to the conv as they are)."""
    lib = _lib.load()
    _f32(gamma, running_var)
    _lib.dev_check(gamma, running_var)
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    _lib.check(lib.tnv3_bn_eval_scale(_lib.ptr(gamma), _lib.ptr(running_var), float(eps), _lib.ptr(scale), c,
                                      _lib.stream_ptr(gamma)))
    return scale


def wino_supported(cin, cout, h, w):
    return bool(_lib.load().tnv3_conv3x3_wino_supported(int(cin), int(cout), int(h), int(w)))


def wino_variant(variant, cin, cout):
    """The kernel variant a call with `variant` (None: tuning.WINO_VARIANT) runs for a cin -> cout layer: -1 is resolved by the
    library from the CHANNEL counts only (tnv3_conv3x3_wino_pick), so the panel packed for a layer fits every image size."""
    if variant is None:
        from . import tuning
        variant = tuning.WINO_VARIANT
    variant = int(variant)
    return int(_lib.load().tnv3_conv3x3_wino_pick(int(cin), int(cout))) if variant < 0 else variant


def wino_layout(variant, cin, cout):
    """Filter pack layout the Winograd kernel `variant` (None / -1: the default for cin -> cout) reads."""
    return int(_lib.load().tnv3_conv3x3_wino_layout(wino_variant(variant, cin, cout)))


def wino_variant_has_stats(variant=None):
    """Whether the Winograd kernel `variant` (None: tuning.WINO_VARIANT; -1 = the library's defaults -- both of them, 5 and 6) can
    emit the BatchNorm batch statistics from its epilogue."""
    if variant is None:
        from . import tuning
        variant = tuning.WINO_VARIANT
    lib = _lib.load()
    if int(variant) < 0:
        return all(bool(lib.tnv3_conv3x3_wino_has_stats(int(lib.tnv3_conv3x3_wino_pick(ci, co)))) for ci, co in ((8, 64), (64, 64), (128, 128)))
    return bool(lib.tnv3_conv3x3_wino_has_stats(int(variant)))


def pack_wino_weights(weight, c_from=0, c_count=None, transpose_flip=False, variant=None):
    """nn.Conv2d weight (Cout, Cin, 3, 3) -> Winograd-domain filters G w G^T for tnv3_conv3x3_wino_forward, of its input
    channels c_from .. c_from + c_count - 1 (default: all).  transpose_flip: the data gradient's filter
    w'[ci][co][kh][kw] = w[co][c_from + ci][2-kh][2-kw] instead (the packed panel then maps Cout -> c_count channels).
    variant: the kernel the panel is for (its layout follows; None: tuning.WINO_VARIANT)."""
    lib = _lib.load()
    _f32(weight)
    weight = weight.contiguous()
    _lib.dev_check(weight)
    cout_w, cin_w = int(weight.shape[0]), int(weight.shape[1])
    c_count = cin_w - int(c_from) if c_count is None else int(c_count)
    cout, cin = (c_count, cout_w) if transpose_flip else (cout_w, c_count)
    layout = wino_layout(variant, cin, cout)
    u = torch.empty(lib.tnv3_conv3x3_wino_packed_floats(cin, cout), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_conv3x3_wino_pack_view(_lib.ptr(weight), _lib.ptr(u), cout_w, cin_w, int(c_from), c_count,
                                               int(bool(transpose_flip)), layout, _lib.stream_ptr(weight)))
    return u


def wino43_supported(cin, cout, h, w):
    return bool(_lib.load().tnv3_conv3x3_wino43_supported(int(cin), int(cout), int(h), int(w)))


def wino43_variant(variant=None):
    """Kernel variant of the F(4x4, 3x3) entries (None: tuning.WINO43_VARIANT): 0 = 16x16x4 MFMAs, all 36 transform coefficients of a block in
    one wave (128-channel workgroups where Cout % 128 == 0), 2 = the same with 64-channel workgroups always; both read one panel layout.
    (1, the 32x32x2 predecessor with its own layout, is dispatched by the diagnostics library only since ABI 6.)"""
    from . import tuning
    return int(tuning.WINO43_VARIANT if variant is None else variant)


def pack_wino43_weights(weight, c_from=0, transpose_flip=False, variant=None):
    """nn.Conv2d weight (Cout, Cin, 3, 3) -> Winograd F(4x4, 3x3) filter panel of its input channels c_from.. (transpose_flip: the
    data gradient's filter) for conv3x3_wino43 (same variant)."""
    lib = _lib.load()
    variant = wino43_variant(variant)
    _f32(weight)
    weight = weight.contiguous()
    _lib.dev_check(weight)
    cout_w, cin_w = int(weight.shape[0]), int(weight.shape[1])
    c_count = cin_w - int(c_from)
    cout, cin = (c_count, cout_w) if transpose_flip else (cout_w, c_count)
    u = torch.empty(lib.tnv3_conv3x3_wino43_packed_floats(cin, cout, variant), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_conv3x3_wino43_pack(_lib.ptr(weight), _lib.ptr(u), cout_w, cin_w, int(c_from), c_count, int(bool(transpose_flip)),
                                            variant, _lib.stream_ptr(weight)))
    return u


def conv3x3_wino43(src, u, cout, mean=None, scale=None, shift=None, relu=False, addend=None, variant=None, pool=False):
    """The plain layer in Winograd F(4x4, 3x3) form (tnv3_conv3x3_wino43_forward): act(((conv3x3(src) + addend) - mean) * scale + shift).
    pool: returns (out, maxpool2x2(out)) -- the pooled tensor from the kernel's write-out (variants 0 / 2; variant 1: a separate pass)."""
    lib = _lib.load()
    variant = wino43_variant(variant)
    _f32(src, u, mean, scale, shift, addend)
    _lib.dev_check(src, u, mean, scale, shift, addend)
    n, cin, h, w = (int(v) for v in src.shape)
    if u.numel() != lib.tnv3_conv3x3_wino43_packed_floats(cin, int(cout), variant):
        raise _lib.Tnv3Error("conv3x3_wino43: filter panel does not match the channel counts")
    out = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=src.device)
    if addend is not None and tuple(addend.shape) != tuple(out.shape):
        raise _lib.Tnv3Error("conv3x3_wino43: addend must have the output's shape")
    fused = pool and variant != 1
    pooled = torch.empty((n, int(cout), h // 2, w // 2), dtype=torch.float32, device=src.device) if fused else None
    if n:
        _lib.check(lib.tnv3_conv3x3_wino43_forward(_lib.ptr(src), _lib.ptr(u), _lib.ptr(addend), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(shift),
                                                   _lib.ptr(out), _lib.ptr(pooled), n, cin, int(cout), h, w, int(bool(relu)), variant, _lib.stream_ptr(src)))
    if pool:
        return out, (pooled if fused else maxpool2x2(out))
    return out


def conv3x3_wino43_stats(src, u, cout, addend=None, variant=None):
    """Training-mode forward of a plain layer in F(4x4, 3x3) form: (z, tile_stats) -- the raw convolution (+ addend) and the per-channel /
    per-tile sums and sums of squares BatchNorm needs, from the same kernel's epilogue (tnv3_conv3x3_wino43_forward_stats)."""
    lib = _lib.load()
    _f32(src, u, addend)
    _lib.dev_check(src, u, addend)
    n, cin, h, w = (int(v) for v in src.shape)
    variant = wino43_variant(variant)
    tiles = int(lib.tnv3_conv3x3_wino43_stats_tiles(n, h, w, variant))
    if tiles <= 0 or u.numel() != lib.tnv3_conv3x3_wino43_packed_floats(cin, int(cout), variant):
        raise _lib.Tnv3Error("conv3x3_wino43_stats: unsupported shape or filter panel mismatch")
    out = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=src.device)
    stats = torch.empty((int(cout), tiles, 2), dtype=torch.float64, device=src.device)
    if addend is not None and tuple(addend.shape) != tuple(out.shape):
        raise _lib.Tnv3Error("conv3x3_wino43_stats: addend must have the output's shape")
    _lib.check(lib.tnv3_conv3x3_wino43_forward_stats(_lib.ptr(src), _lib.ptr(u), _lib.ptr(addend), _lib.ptr(out), _lib.ptr(stats), n, cin,
                                                     int(cout), h, w, variant, _lib.stream_ptr(src)))
    return out, stats


class _WinoPackItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("u", ctypes.c_void_p), ("cout_w", ctypes.c_int), ("cin_w", ctypes.c_int), ("c_from", ctypes.c_int),
                ("c_count", ctypes.c_int), ("transpose_flip", ctypes.c_int), ("layout", ctypes.c_int)]


def pack_wino_weights_multi(specs, variant=None):
    """[pack_wino_weights(weight, c_from=c_from, transpose_flip=flip) for (weight, c_from, flip) in specs] in ONE launch
    (tnv3_conv3x3_wino_pack_multi): the panels a training step rebuilds after every optimiser step.  A fourth element 43 in a spec
    asks for the F(4x4, 3x3) panel (pack_wino43_weights) instead; a fifth names the pack layout (0-4) explicitly instead of deriving it
    from the tuning switches.  All weights on one device; ordered on that device's current
    stream.  Bit-identical to the one-panel calls."""
    lib = _lib.load()
    if not specs:
        return []
    items = (_WinoPackItem * len(specs))()
    outs, keep = [], []
    dev = specs[0][0].device
    for k, spec in enumerate(specs):
        weight, c_from, flip = spec[:3]
        f43 = len(spec) > 3 and spec[3] == 43
        _f32(weight)
        if weight.device != dev:
            raise _lib.Tnv3Error("pack_wino_weights_multi: all weights must live on one device")
        weight = weight.contiguous()
        keep.append(weight)
        cout_w, cin_w = int(weight.shape[0]), int(weight.shape[1])
        c_count = cin_w - int(c_from)
        cout, cin = (c_count, cout_w) if flip else (cout_w, c_count)
        layout = int(spec[4]) if len(spec) > 4 else None
        v43 = (1 if layout == 3 else 0) if (f43 and layout is not None) else wino43_variant(None)
        floats = lib.tnv3_conv3x3_wino43_packed_floats(cin, cout, v43) if f43 else lib.tnv3_conv3x3_wino_packed_floats(cin, cout)
        u = torch.empty(floats, dtype=torch.float32, device=dev)
        outs.append(u)
        items[k] = _WinoPackItem(_lib.ptr(weight), _lib.ptr(u), cout_w, cin_w, int(c_from), c_count, int(bool(flip)),
                                 layout if layout is not None else ((3 if v43 == 1 else 4) if f43 else wino_layout(variant, cin, cout)))
    _lib.dev_check(keep[0])
    _lib.check(lib.tnv3_conv3x3_wino_pack_multi(ctypes.cast(items, ctypes.c_void_p), len(specs), _lib.stream_ptr(keep[0])))
    return outs


def conv3x3_wino(src, u, cout, mean=None, scale=None, shift=None, relu=False, addend=None, variant=None):
    """The plain eval-mode layer in Winograd F(2x2, 3x3) form (tnv3_conv3x3_wino_forward).  variant: kernel family for THIS
    call (None: tuning.WINO_VARIANT; -1 = the library's pick: 6, the 128-channel form, where Cout % 128 == 0, else 5, the streaming persistent
    kernel.  The earlier generations 0 / 2 / 3 / 4 / 7 are measurement twins of the diagnostics library)."""
    lib = _lib.load()
    _f32(src, u, mean, scale, shift, addend)
    _lib.dev_check(src, u, mean, scale, shift, addend)
    n, cin, h, w = (int(v) for v in src.shape)
    if u.numel() != lib.tnv3_conv3x3_wino_packed_floats(cin, int(cout)):
        raise _lib.Tnv3Error("conv3x3_wino: transformed-filter buffer does not match the channel counts")
    out = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=src.device)
    if addend is not None and tuple(addend.shape) != tuple(out.shape):
        raise _lib.Tnv3Error("conv3x3_wino: addend must have the output's shape")
    if n:
        variant = wino_variant(variant, cin, cout)
        _lib.check(lib.tnv3_conv3x3_wino_forward(_lib.ptr(src), _lib.ptr(u), _lib.ptr(addend), _lib.ptr(mean), _lib.ptr(scale),
                                                 _lib.ptr(shift), _lib.ptr(out), n, cin, int(cout), h, w, int(bool(relu)),
                                                 int(variant), _lib.stream_ptr(src)))
    return out


def conv3x3_wino_stats(src, u, cout, addend=None, variant=None):
    """Training-mode forward of a plain layer: (z, tile_stats) -- the raw convolution (+ addend) and, from the same kernel's
    epilogue, the per-channel / per-tile sums and sums of squares BatchNorm needs (tnv3_conv3x3_wino_forward_stats)."""
    lib = _lib.load()
    _f32(src, u, addend)
    _lib.dev_check(src, u, addend)
    n, cin, h, w = (int(v) for v in src.shape)
    variant = wino_variant(variant, cin, cout)
    tiles = int(lib.tnv3_conv3x3_wino_stats_tiles(n, h, w, variant))
    if tiles <= 0 or u.numel() != lib.tnv3_conv3x3_wino_packed_floats(cin, int(cout)):
        raise _lib.Tnv3Error("conv3x3_wino_stats: unsupported shape or filter buffer mismatch")
    out = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=src.device)
    stats = torch.empty((int(cout), tiles, 2), dtype=torch.float64, device=src.device)
    if addend is not None and tuple(addend.shape) != tuple(out.shape):
        raise _lib.Tnv3Error("conv3x3_wino_stats: addend must have the output's shape")
    _lib.check(lib.tnv3_conv3x3_wino_forward_stats(_lib.ptr(src), _lib.ptr(u), _lib.ptr(addend), _lib.ptr(out), _lib.ptr(stats), n, cin,
                                                   int(cout), h, w, int(variant), _lib.stream_ptr(src)))
    return out, stats


def pack_up2x_weights(weight, c0):
    """nn.Conv2d weight (Cout, Cin, 3, 3) -> pre-summed class filters of its first c0 (upsampled) input channels."""
    lib = _lib.load()
    _f32(weight)
    _lib.dev_check(weight)
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    wq = torch.empty(lib.tnv3_conv_up2x_packed_floats(int(c0), cout), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_pack_up2x_weights(_lib.ptr(weight), _lib.ptr(wq), cout, cin, int(c0), _lib.stream_ptr(weight)))
    return wq


def conv_up2x(src_low, wq, cout, cfg=-1):
    """Partial sums of conv3x3 over the nearest-2x upsampling of src_low, computed at the low resolution (tnv3_conv_up2x_forward)."""
    lib = _lib.load()
    _f32(src_low, wq)
    _lib.dev_check(src_low, wq)
    n, c0, hl, wl = (int(v) for v in src_low.shape)
    if wq.numel() != lib.tnv3_conv_up2x_packed_floats(c0, int(cout)):
        raise _lib.Tnv3Error("conv_up2x: class-filter buffer does not match the channel counts")
    out = torch.empty((n, int(cout), 2 * hl, 2 * wl), dtype=torch.float32, device=src_low.device)
    if n:
        _lib.check(lib.tnv3_conv_up2x_forward(_lib.ptr(src_low), _lib.ptr(wq), _lib.ptr(out), n, c0, int(cout), hl, wl, int(cfg),
                                              _lib.stream_ptr(src_low)))
    return out


def up2x_wino_variant(variant=None):
    """Kernel of the upsampled half's forward (None: tuning.UP2X_WINO_VARIANT): 0 / 1 the 9-GEMM F(2x2) kernel, 2 the 25-product F(4x4) form on
    the 16x16x4 kernel.  0 / 1 and 2 read different panels: pack and run with the same one."""
    from . import tuning
    v = int(tuning.UP2X_WINO_VARIANT if variant is None else variant)
    return 0 if v < 0 else v


def up2x_wino_supported(c0, cout, hl, wl, variant=None):
    return bool(_lib.load().tnv3_conv_up2x_wino_supported(int(c0), int(cout), int(hl), int(wl), up2x_wino_variant(variant)))


def pack_up2x_wino_weights(weight, c0, variant=None):
    """U' of the first c0 (upsampled) input channels of a decoder-entry layer's weight (tnv3_conv_up2x_wino_pack)."""
    lib = _lib.load()
    _f32(weight)
    weight = weight.contiguous()
    _lib.dev_check(weight)
    variant = up2x_wino_variant(variant)
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    u = torch.empty(lib.tnv3_conv_up2x_wino_packed_floats(int(c0), cout, 2 if variant == 2 else 0), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_conv_up2x_wino_pack(_lib.ptr(weight), _lib.ptr(u), cout, cin, int(c0), 2 if variant == 2 else 0, _lib.stream_ptr(weight)))
    return u


def conv_up2x_wino(src_low, u, cout, variant=None):
    """Partial sums of conv3x3 over the nearest-2x upsampling of src_low in Winograd form (tnv3_conv_up2x_wino_forward): 9 of the 16 F(2x2)
    GEMMs (variant 0; 1 = the other wave group's MFMAs first) or 25 of the 36 F(4x4) products (variant 2)."""
    lib = _lib.load()
    _f32(src_low, u)
    _lib.dev_check(src_low, u)
    variant = up2x_wino_variant(variant)
    n, c0, hl, wl = (int(v) for v in src_low.shape)
    if u.numel() != lib.tnv3_conv_up2x_wino_packed_floats(c0, int(cout), 2 if variant == 2 else 0):
        raise _lib.Tnv3Error("conv_up2x_wino: filter buffer does not match the channel counts")
    out = torch.empty((n, int(cout), 2 * hl, 2 * wl), dtype=torch.float32, device=src_low.device)
    if n:
        _lib.check(lib.tnv3_conv_up2x_wino_forward(_lib.ptr(src_low), _lib.ptr(u), _lib.ptr(out), n, c0, int(cout), hl, wl,
                                                   int(variant), _lib.stream_ptr(src_low)))
    return out


def dgrad_up2x_wino_variant(variant=None):
    """Kernel of the upsampled half's data gradient (None: tuning.DGRAD_UP2X_WINO_VARIANT): 0 / 1 the one-GEMM F(2x2) kernel (K = 9 * Cout),
    2 the 25-product F(4x4) form on the 16x16x4 kernel (MODE 2 of kernels/conv3x3_wino43s_mfma.h; its own panel)."""
    from . import tuning
    v = int(tuning.DGRAD_UP2X_WINO_VARIANT if variant is None else variant)
    return 0 if v < 0 else v


def dgrad_up2x_wino_supported(c0, cout, hl, wl, variant=None):
    return bool(_lib.load().tnv3_dgrad_up2x_wino_supported(int(c0), int(cout), int(hl), int(wl), dgrad_up2x_wino_variant(variant)))


def pack_dgrad_up2x_wino_weights(weight, c0, variant=None):
    """U'' of the first c0 (upsampled) input channels for the low-resolution data gradient (tnv3_dgrad_up2x_wino_pack); the panel follows
    the kernel variant."""
    lib = _lib.load()
    _f32(weight)
    _lib.dev_check(weight)
    v = 2 if dgrad_up2x_wino_variant(variant) == 2 else 0
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    u = torch.empty(lib.tnv3_dgrad_up2x_wino_packed_floats(int(c0), cout, v), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_dgrad_up2x_wino_pack(_lib.ptr(weight), _lib.ptr(u), cout, cin, int(c0), v, _lib.stream_ptr(weight)))
    return u


def dgrad_up2x_wino(dz, u, c0, variant=None):
    """Gradient w.r.t. the low-res operand of nn.Upsample(2) -> conv3x3 at the low resolution (tnv3_dgrad_up2x_wino): variant 0 / 1 as one
    GEMM with K = 9 * Cout, 2 in the 25-of-36 F(4x4) form; `u` must be the panel packed for that variant."""
    lib = _lib.load()
    _f32(dz, u)
    _lib.dev_check(dz, u)
    v = dgrad_up2x_wino_variant(variant)
    n, cout, h, w = (int(v_) for v_ in dz.shape)
    if (h | w) & 1 or u.numel() != lib.tnv3_dgrad_up2x_wino_packed_floats(int(c0), cout, 2 if v == 2 else 0):
        raise _lib.Tnv3Error("dgrad_up2x_wino: odd output size or filter buffer / channel mismatch (a panel of the other variant?)")
    out = torch.empty((n, int(c0), h // 2, w // 2), dtype=torch.float32, device=dz.device)
    if n:
        _lib.check(lib.tnv3_dgrad_up2x_wino(_lib.ptr(dz), _lib.ptr(u), _lib.ptr(out), n, int(c0), cout, h // 2, w // 2, v, _lib.stream_ptr(dz)))
    return out


def dgrad_up2x_wino_bnstats(dz, u, c0, z, mean, invstd, gamma, beta):
    """dgrad_up2x_wino(variant 2) that also takes the two BatchNorm + ReLU backward sums of the block whose activation is upsampled (z: its raw
    output at the low resolution): returns (d_low, tile_stats [C0][N * (H_low / 2) * (W_low / 32)][2] float64) for bn_relu_backward_tiles."""
    lib = _lib.load()
    _f32(dz, u, z, mean, invstd, gamma, beta)
    _lib.dev_check(dz, u, z, mean, invstd, gamma, beta)
    n, cout, h, w = (int(v_) for v_ in dz.shape)
    c0 = int(c0)
    if (h | w) & 1 or u.numel() != lib.tnv3_dgrad_up2x_wino_packed_floats(c0, cout, 2):
        raise _lib.Tnv3Error("dgrad_up2x_wino_bnstats: odd output size or filter buffer / channel mismatch (the panel of variant 2 is required)")
    hl, wl = h // 2, w // 2
    if tuple(z.shape) != (n, c0, hl, wl) or not z.is_contiguous() or not dz.is_contiguous():
        raise _lib.Tnv3Error("dgrad_up2x_wino_bnstats: z must be the contiguous (N, C0, H / 2, W / 2) raw output of the upsampled block")
    if not lib.tnv3_dgrad_up2x_wino_supported(c0, cout, hl, wl, 2):
        raise _lib.Tnv3Error(f"dgrad_up2x_wino_bnstats: unsupported shape {c0} <- {cout}, {hl}x{wl}")
    out = torch.empty((n, c0, hl, wl), dtype=torch.float32, device=dz.device)
    st = torch.empty((c0, n * (hl // 2) * (wl // 32), 2), dtype=torch.float64, device=dz.device)
    _lib.check(lib.tnv3_dgrad_up2x_wino_bnstats(_lib.ptr(dz), _lib.ptr(u), _lib.ptr(out), _lib.ptr(st), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(invstd),
                                                _lib.ptr(gamma), _lib.ptr(beta), n, c0, cout, hl, wl, 2, _lib.stream_ptr(dz)))
    return out, st


def pack_dgrad_up2x_weights(weight, c0):
    """nn.Conv2d weight (Cout, Cin, 3, 3) -> the 4x4 stride-2 filters of the low-resolution data gradient (first c0 inputs)."""
    lib = _lib.load()
    _f32(weight)
    _lib.dev_check(weight)
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    g = torch.empty(lib.tnv3_dgrad_up2x_packed_floats(int(c0), cout), dtype=torch.float32, device=weight.device)
    _lib.check(lib.tnv3_pack_dgrad_up2x_weights(_lib.ptr(weight), _lib.ptr(g), cout, cin, int(c0), _lib.stream_ptr(weight)))
    return g


def dgrad_up2x(dz, g, c0):
    """Gradient w.r.t. the low-res operand of nn.Upsample(2) -> conv3x3, straight at the low resolution (tnv3_dgrad_up2x)."""
    lib = _lib.load()
    _f32(dz, g)
    _lib.dev_check(dz, g)
    n, cout, h, w = (int(v) for v in dz.shape)
    if (h | w) & 1 or g.numel() != lib.tnv3_dgrad_up2x_packed_floats(int(c0), cout):
        raise _lib.Tnv3Error("dgrad_up2x: odd output size or filter buffer / channel mismatch")
    out = torch.empty((n, int(c0), h // 2, w // 2), dtype=torch.float32, device=dz.device)
    if n:
        _lib.check(lib.tnv3_dgrad_up2x(_lib.ptr(dz), _lib.ptr(g), _lib.ptr(out), n, int(c0), cout, h // 2, w // 2, _lib.stream_ptr(dz)))
    return out


def conv3x3(src0, wpack, cout, src1=None, mean=None, scale=None, shift=None, up0=False, relu=False, cfg=-1, out=None, addend=None):
    """act((conv3x3(cat([up2x?(src0), src1], 1), W) + addend - mean) * scale + shift) -- see tnv3_conv3x3_forward(_add)."""
    lib = _lib.load()
    _f32(src0, src1, wpack, mean, scale, shift, out, addend)
    _lib.dev_check(src0, src1, wpack, mean, scale, shift, out, addend)
    n, c0, h0, w0 = (int(v) for v in src0.shape)
    h, w = (2 * h0, 2 * w0) if up0 else (h0, w0)
    c1 = 0
    if src1 is not None:
        if int(src1.shape[0]) != n or tuple(src1.shape[2:]) != (h, w):
            raise _lib.Tnv3Error(f"conv3x3: src1 shape {tuple(src1.shape)} does not match ({n}, *, {h}, {w})")
        c1 = int(src1.shape[1])
    need = lib.tnv3_conv3x3_packed_floats(cout, c0 + c1, 0)
    if wpack.numel() != need:
        raise _lib.Tnv3Error(f"conv3x3: packed filter has {wpack.numel()} floats, expected {need}")
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=src0.device)
    elif tuple(out.shape) != (n, cout, h, w):
        raise _lib.Tnv3Error("conv3x3: wrong output shape")
    if n == 0:
        return out                                   # empty batch: nothing to launch (torch ops accept N = 0 too)
    if addend is not None and tuple(addend.shape) != (n, cout, h, w):
        raise _lib.Tnv3Error("conv3x3: addend must have the output's shape")
    _lib.check(lib.tnv3_conv3x3_forward_add(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(wpack), _lib.ptr(addend), _lib.ptr(mean),
                                            _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(out), n, c0, c1, cout, h, w, int(bool(up0)),
                                            int(bool(relu)), int(cfg), _lib.stream_ptr(src0)))
    return out


def head1x1_sigmoid(x, weight, bias, apply_sigmoid=True, out=None):
    """sigmoid(conv1x1(x) + b): weight (L, C, 1, 1) or (L, C)."""
    lib = _lib.load()
    _f32(x, weight, bias, out)
    _lib.dev_check(x, weight, bias, out)
    n, c, h, w = (int(v) for v in x.shape)
    l = int(weight.shape[0])
    if weight.numel() != l * c or bias.numel() != l:
        raise _lib.Tnv3Error("head1x1: weight/bias shape mismatch")
    if out is None:
        out = torch.empty((n, l, h, w), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    _lib.check(lib.tnv3_head1x1_sigmoid(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), n, c, l, h * w,
                                        int(bool(apply_sigmoid)), _lib.stream_ptr(x)))
    return out


def maxpool2x2(x, out=None):
    lib = _lib.load()
    _f32(x, out)
    _lib.dev_check(x, out)
    n, c, h, w = (int(v) for v in x.shape)
    if out is None:
        out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    _lib.check(lib.tnv3_maxpool2x2(_lib.ptr(x), _lib.ptr(out), n * c, h, w, _lib.stream_ptr(x)))
    return out


ACT_NONE, ACT_LEAKY_RELU, ACT_SIGMOID = 0, 1, 2


def conv1d_k3(src0, weight, bias, src1=None, src_nlc=False, dst_nlc=False, act=ACT_LEAKY_RELU):
    """act(conv1d_k3_same(cat([src0, src1], channels), weight) + bias) -- see tnv3_conv1d_k3_forward."""
    lib = _lib.load()
    _f32(src0, src1, weight, bias)
    _lib.dev_check(src0, src1, weight, bias)
    if src_nlc:
        n, l, c0 = (int(v) for v in src0.shape)
        c1 = int(src1.shape[2]) if src1 is not None else 0
    else:
        n, c0, l = (int(v) for v in src0.shape)
        c1 = int(src1.shape[1]) if src1 is not None else 0
    cout = int(weight.shape[0])
    if tuple(weight.shape) != (cout, c0 + c1, 3) or bias.numel() != cout:
        raise _lib.Tnv3Error(f"conv1d_k3: weight {tuple(weight.shape)} does not match {c0}+{c1} input channels")
    out = torch.empty((n, l, cout) if dst_nlc else (n, cout, l), dtype=torch.float32, device=src0.device)
    if n == 0:
        return out
    _lib.check(lib.tnv3_conv1d_k3_forward(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out),
                                          n, c0, c1, cout, l, int(bool(src_nlc)), int(bool(dst_nlc)), int(act),
                                          _lib.stream_ptr(src0)))
    return out


def inpaintnet_pack(weights, biases):
    """The nine (weight, bias) pairs of an InpaintNet in network order -> the packed parameter buffer of the fused kernel."""
    lib = _lib.load()
    if len(weights) != 9 or len(biases) != 9:
        raise _lib.Tnv3Error("inpaintnet_pack: expected nine layers")
    ws = [w.contiguous() for w in weights]
    bs = [b.contiguous() for b in biases]
    _f32(*ws, *bs)
    _lib.dev_check(*ws, *bs)
    shapes = [(32, 3, 3), (64, 32, 3), (128, 64, 3), (256, 128, 3), (256, 256, 3), (128, 384, 3), (64, 192, 3), (32, 96, 3), (2, 32, 3)]
    for w, b, sh in zip(ws, bs, shapes):
        if tuple(w.shape) != sh or b.numel() != sh[0]:
            raise _lib.Tnv3Error(f"inpaintnet_pack: expected a weight of shape {sh}, got {tuple(w.shape)}")
    packed = torch.empty(lib.tnv3_inpaintnet_packed_floats(), dtype=torch.float32, device=ws[0].device)
    _lib.check(lib.tnv3_inpaintnet_pack(_ptr_array(ws), _ptr_array(bs), _lib.ptr(packed), _lib.stream_ptr(packed)))
    return packed


def inpaintnet_fused(x, m, packed):
    """InpaintNet.forward as one kernel (tnv3_inpaintnet_fused_forward): x (N, 16, 2), m (N, 16, 1) -> (N, 16, 2)."""
    lib = _lib.load()
    _f32(x, m, packed)
    _lib.dev_check(x, m, packed)
    n, l = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((n, l, 2), dtype=torch.float32, device=x.device)
    if n:
        _lib.check(lib.tnv3_inpaintnet_fused_forward(_lib.ptr(x), _lib.ptr(m), _lib.ptr(packed), _lib.ptr(out), n, l, _lib.stream_ptr(x)))
    return out


def inpaintnet_pack_t(weights):
    """Transposed, tap-flipped filters of the seven dense InpaintNet layers in lane order (tnv3_inpaintnet_pack_t), for the fused backward."""
    lib = _lib.load()
    if len(weights) != 9:
        raise _lib.Tnv3Error("inpaintnet_pack_t: expected nine layers")
    ws = [w.contiguous() for w in weights]
    _f32(*ws)
    _lib.dev_check(*ws)
    packed_t = torch.empty(lib.tnv3_inpaintnet_packed_t_floats(), dtype=torch.float32, device=ws[0].device)
    _lib.check(lib.tnv3_inpaintnet_pack_t(_ptr_array(ws), _lib.ptr(packed_t), _lib.stream_ptr(packed_t)))
    return packed_t


def inpaintnet_fused_train_forward(x, m, packed):
    """The fused forward that also saves the hidden activations: (out (N, 16, 2), acts (N, 960, 16))."""
    lib = _lib.load()
    _f32(x, m, packed)
    _lib.dev_check(x, m, packed)
    n, l = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((n, l, 2), dtype=torch.float32, device=x.device)
    acts = torch.empty(lib.tnv3_inpaintnet_act_floats(max(n, 1)), dtype=torch.float32, device=x.device)
    if n:
        _lib.check(lib.tnv3_inpaintnet_fused_forward_train(_lib.ptr(x), _lib.ptr(m), _lib.ptr(packed), _lib.ptr(out), _lib.ptr(acts), n, l,
                                                           _lib.stream_ptr(x)))
    return out, acts


def inpaintnet_fused_backward(x, m, dout, out, acts, packed, packed_t):
    """All 18 parameter gradients of an InpaintNet step as ONE flat tensor in state_dict order (tnv3_inpaintnet_fused_backward)."""
    lib = _lib.load()
    _f32(x, m, dout, out, acts, packed, packed_t)
    _lib.dev_check(x, m, dout, out, acts, packed, packed_t)
    n, l = int(x.shape[0]), int(x.shape[1])
    grads = torch.empty(lib.tnv3_inpaintnet_param_floats(), dtype=torch.float32, device=x.device)
    if n == 0:
        return grads.zero_()
    dpre = torch.empty(lib.tnv3_inpaintnet_dpre_floats(n), dtype=torch.float32, device=x.device)
    _lib.check(lib.tnv3_inpaintnet_fused_backward(_lib.ptr(x), _lib.ptr(m), _lib.ptr(dout), _lib.ptr(out), _lib.ptr(acts), _lib.ptr(packed),
                                                  _lib.ptr(packed_t), _lib.ptr(dpre), _lib.ptr(grads), n, l, _lib.stream_ptr(x)))
    return grads


def ensemble_frames(win, s_base, weight, t0, n_frames, num_sample, sum_order=None):
    """Temporal ensemble of global frames t0..t0+n_frames-1 from resident windows win[i] = window s_base+i.
    win: (n_local, L, *tail) -> out: (n_frames, *tail).  See tnv3_ensemble_frames.  sum_order (default: by the tail size, as
    torch's CPU sum kernel picks its path): 0 = rows added sequentially (heat maps), 1 = four interleaved partial sums (fewer
    than four elements per position: the (L, 2) coordinates) -- bit-identical to predict.py's `.sum(0)` either way."""
    lib = _lib.load()
    _f32(win, weight)
    _lib.dev_check(win, weight)
    n_local, l = int(win.shape[0]), int(win.shape[1])
    tail = tuple(win.shape[2:])
    e = 1
    for v in tail:
        e *= int(v)
    out = torch.empty((n_frames,) + tail, dtype=torch.float32, device=win.device)
    if n_frames == 0:
        return out
    if sum_order is None:
        sum_order = 1 if e < 4 else 0
    _lib.check(lib.tnv3_ensemble_frames(_lib.ptr(win), n_local, int(s_base), l, e, _lib.ptr(weight), int(t0), int(n_frames),
                                        int(num_sample), int(sum_order), _lib.ptr(out), _lib.stream_ptr(win)))
    return out


def heatmap_peakfind(heat, threshold=0.5, tie_last_wins=True):
    """(frames, H, W) fp32 heat maps -> (frames, 4) int32 boxes (x, y, w, h); zeros for an empty map."""
    lib = _lib.load()
    _f32(heat)
    _lib.dev_check(heat)
    if heat.dim() != 3:
        raise _lib.Tnv3Error("heatmap_peakfind: expected (frames, H, W)")
    frames, h, w = (int(v) for v in heat.shape)
    out = torch.empty((frames, 4), dtype=torch.int32, device=heat.device)
    if frames == 0:
        return out
    step = 4096
    for f0 in range(0, frames, step):
        nf = min(step, frames - f0)
        nbytes = lib.tnv3_peakfind_workspace_bytes(nf, h, w)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=heat.device)
        _lib.check(lib.tnv3_heatmap_peakfind(_lib.ptr(heat[f0:f0 + nf]), float(threshold), int(bool(tie_last_wins)),
                                             _lib.ptr(out[f0:f0 + nf]), _lib.ptr(ws), ws.numel() * 8, nf, h, w,
                                             _lib.stream_ptr(heat)))
    return out


def heatmap_box_max(heat, boxes=None):
    """(frames, H, W) fp32 maps [+ (frames, 4) int32 boxes (x, y, w, h)] -> (frames,) maxima (0 for an empty box)."""
    lib = _lib.load()
    _f32(heat)
    _lib.dev_check(heat, boxes)
    if heat.dim() != 3:
        raise _lib.Tnv3Error("heatmap_box_max: expected (frames, H, W)")
    frames, h, w = (int(v) for v in heat.shape)
    if boxes is not None and (boxes.dtype != torch.int32 or tuple(boxes.shape) != (frames, 4) or not boxes.is_contiguous()):
        raise _lib.Tnv3Error("heatmap_box_max: boxes must be contiguous int32 (frames, 4)")
    out = torch.empty(frames, dtype=torch.float32, device=heat.device)
    if frames:
        _lib.check(lib.tnv3_heatmap_box_max(_lib.ptr(heat), _lib.ptr(boxes), _lib.ptr(out), frames, h, w, _lib.stream_ptr(heat)))
    return out


# ------------------------------------------------------------------------------------------------- training ops
def _workspace(nbytes, device):
    return torch.empty((int(nbytes) + 7) // 8, dtype=torch.int64, device=device)


def _grad_out(shape, device, out, what):
    """The tensor a gradient kernel writes: `out` when the caller names a destination (the data-parallel reducer hands out views of its
    flat all-reduce buckets, so a gradient is born where the collective reads it -- no copy), else a fresh one.  A destination must be a
    contiguous fp32 tensor of the gradient's element count on the gradient's device, 16-byte aligned (some folds store float4)."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    n = 1
    for v in shape:
        n *= int(v)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n or out.device != device or out.data_ptr() % 16:
        raise _lib.Tnv3Error(f"{what}: `out` must be a contiguous, 16-byte aligned fp32 tensor of {n} elements on {device}")
    return out.view(shape)


def bn_train_forward(z, gamma, beta, running_mean, running_var, eps=BN_EPS, momentum=0.1, tile_stats=None, pool=False):
    """Training-mode BatchNorm2d + ReLU on the raw conv output; updates running stats in place.
    Returns (a, save_mean, save_invstd).  tile_stats: the (C, tiles, 2) float64 sums the producing convolution's epilogue left
    (conv3x3_wino_stats) -- the statistics pass over z is then skipped.  pool=True: returns (a, save_mean, save_invstd, MaxPool2d(2, 2)(a)) --
    written by the normalise + ReLU pass itself where the shape allows (tile_stats given, H % 2 == 0, W % 4 == 0), else by maxpool2x2."""
    if pool:
        n, c, h, w = (int(v) for v in z.shape)
        if tile_stats is None or (h & 1) or (w & 3):
            a, mean, invstd = bn_train_forward(z, gamma, beta, running_mean, running_var, eps, momentum, tile_stats)
            return a, mean, invstd, maxpool2x2(a)
        lib = _lib.load()
        _f32(z, gamma, beta, running_mean, running_var)
        _lib.dev_check(z, gamma, beta, running_mean, running_var, tile_stats)
        if tile_stats.dtype != torch.float64 or tile_stats.dim() != 3 or int(tile_stats.shape[0]) != c or int(tile_stats.shape[2]) != 2:
            raise _lib.Tnv3Error("bn_train_forward: tile_stats must be float64 (C, tiles, 2)")
        a = torch.empty_like(z)
        pooled = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=z.device)
        mean = torch.empty(c, dtype=torch.float32, device=z.device)
        invstd = torch.empty_like(mean)
        ws = _workspace(lib.tnv3_bn_workspace_bytes(c), z.device)
        _lib.check(lib.tnv3_bn_train_forward_tiles_pool(_lib.ptr(z), _lib.ptr(tile_stats), int(tile_stats.shape[1]), _lib.ptr(gamma), _lib.ptr(beta),
                                                        _lib.ptr(running_mean), _lib.ptr(running_var), float(eps), float(momentum), _lib.ptr(a),
                                                        _lib.ptr(pooled), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(ws), ws.numel() * 8, n, c, h, w,
                                                        _lib.stream_ptr(z)))
        return a, mean, invstd, pooled
    lib = _lib.load()
    _f32(z, gamma, beta, running_mean, running_var)
    _lib.dev_check(z, gamma, beta, running_mean, running_var, tile_stats)
    n, c, h, w = (int(v) for v in z.shape)
    a = torch.empty_like(z)
    mean = torch.empty(c, dtype=torch.float32, device=z.device)
    invstd = torch.empty_like(mean)
    ws = _workspace(lib.tnv3_bn_workspace_bytes(c), z.device)
    if tile_stats is not None:
        if tile_stats.dtype != torch.float64 or tile_stats.dim() != 3 or int(tile_stats.shape[0]) != c or int(tile_stats.shape[2]) != 2:
            raise _lib.Tnv3Error("bn_train_forward: tile_stats must be float64 (C, tiles, 2)")
        _lib.check(lib.tnv3_bn_train_forward_tiles(_lib.ptr(z), _lib.ptr(tile_stats), int(tile_stats.shape[1]), _lib.ptr(gamma), _lib.ptr(beta),
                                                   _lib.ptr(running_mean), _lib.ptr(running_var), float(eps), float(momentum), _lib.ptr(a),
                                                   _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(ws), ws.numel() * 8, n, c, h * w,
                                                   _lib.stream_ptr(z)))
        return a, mean, invstd
    _lib.check(lib.tnv3_bn_train_forward(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                         _lib.ptr(running_var), float(eps), float(momentum), _lib.ptr(a), _lib.ptr(mean),
                                         _lib.ptr(invstd), _lib.ptr(ws), ws.numel() * 8, n, c, h * w, _lib.stream_ptr(z)))
    return a, mean, invstd


def bn_relu_backward(da, a, z, gamma, mean, invstd, inplace=True, beta=None, out=None):
    """Returns (dz, dgamma, dbeta); dz overwrites da when inplace.  With a=None the ReLU mask is recomputed from z, gamma,
    beta and the saved statistics (bit-identical to a > 0; gamma / beta must be the forward call's values).  out: (dgamma, dbeta)
    destinations (either may be None)."""
    lib = _lib.load()
    if a is None and beta is None:
        raise _lib.Tnv3Error("bn_relu_backward: needs the forward output a or beta")
    _f32(da, z, gamma, mean, invstd, *(t for t in (a, beta) if t is not None))
    _lib.dev_check(da, a, z, gamma, beta, mean, invstd)
    n, c, h, w = (int(v) for v in z.shape)
    dz = da if inplace else torch.empty_like(da)
    dgamma = _grad_out((c,), z.device, out[0] if out else None, "bn_relu_backward")
    dbeta = _grad_out((c,), z.device, out[1] if out else None, "bn_relu_backward")
    ws = _workspace(lib.tnv3_bn_workspace_bytes(c), z.device)
    _lib.check(lib.tnv3_bn_relu_backward(_lib.ptr(da), _lib.ptr(a), _lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd),
                                         _lib.ptr(dz), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel() * 8, n, c,
                                         h * w, _lib.stream_ptr(z)))
    return dz, dgamma, dbeta


def bn_bwd_consts(mean, invstd, gamma, beta):
    """(C, 4) per-channel constants (mean, invstd, gamma * invstd as the passes round it, beta) for conv3x3_wino_dgrad_bnstats."""
    lib = _lib.load()
    _f32(mean, invstd, gamma, beta)
    _lib.dev_check(mean, invstd, gamma, beta)
    c = int(mean.numel())
    c4 = torch.empty((c, 4), dtype=torch.float32, device=mean.device)
    _lib.check(lib.tnv3_bn_bwd_consts(_lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(c4), c, _lib.stream_ptr(mean)))
    return c4


def conv3x3_wino_dgrad_bnstats(dz, u, cout, bn_z, bn_c4, variant=None):
    """The Winograd data gradient dA = conv3x3(dZ, W^T flipped) that ALSO takes, from its epilogue's registers, the two sums of the
    previous block's BatchNorm + ReLU backward: returns (dA, tile_stats (cout, tiles, 2) float64) -- feed them to
    bn_relu_backward_tiles.  bn_z: that block's raw convolution output (the shape of dA), bn_c4: bn_bwd_consts(...) of it."""
    lib = _lib.load()
    _f32(dz, u, bn_z, bn_c4)
    _lib.dev_check(dz, u, bn_z, bn_c4)
    n, cin, h, w = (int(v) for v in dz.shape)
    variant = wino_variant(variant, cin, cout)
    tiles = int(lib.tnv3_conv3x3_wino_stats_tiles(n, h, w, variant))
    if tiles <= 0 or u.numel() != lib.tnv3_conv3x3_wino_packed_floats(cin, int(cout)) or tuple(bn_z.shape) != (n, int(cout), h, w):
        raise _lib.Tnv3Error("conv3x3_wino_dgrad_bnstats: unsupported shape, filter buffer or z mismatch")
    da = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=dz.device)
    stats = torch.empty((int(cout), tiles, 2), dtype=torch.float64, device=dz.device)
    _lib.check(lib.tnv3_conv3x3_wino_dgrad_bnstats(_lib.ptr(dz), _lib.ptr(u), _lib.ptr(da), _lib.ptr(stats), _lib.ptr(bn_z), _lib.ptr(bn_c4), n, cin,
                                                   int(cout), h, w, int(variant), _lib.stream_ptr(dz)))
    return da, stats


def conv3x3_wino43_dgrad_bnstats(dz, u_t, cout, bn_z, bn_mean, bn_invstd, bn_gamma, bn_beta, variant=None):
    """The F(4x4) data gradient dA = conv3x3(dZ, W^T flipped) that ALSO takes, from its write-out's registers, the two sums of the previous
    block's BatchNorm + ReLU backward (tnv3_conv3x3_wino43_dgrad_bnstats, ABI 7): returns (dA, tile_stats (cout, tiles, 2) float64) -- feed
    them to bn_relu_backward_tiles.  bn_z: that block's raw convolution output (the shape of dA); bn_mean / bn_invstd: its saved batch
    statistics; bn_gamma / bn_beta: its affine.  dA is bit-identical to conv3x3_wino43(dz, u_t, cout)."""
    lib = _lib.load()
    _f32(dz, u_t, bn_z, bn_mean, bn_invstd, bn_gamma, bn_beta)
    _lib.dev_check(dz, u_t, bn_z, bn_mean, bn_invstd, bn_gamma, bn_beta)
    n, cin, h, w = (int(v) for v in dz.shape)
    variant = wino43_variant(variant)
    tiles = int(lib.tnv3_conv3x3_wino43_stats_tiles(n, h, w, variant))
    if (tiles <= 0 or variant == 1 or u_t.numel() != lib.tnv3_conv3x3_wino43_packed_floats(cin, int(cout), variant)
            or tuple(bn_z.shape) != (n, int(cout), h, w) or any(int(t.numel()) != int(cout) for t in (bn_mean, bn_invstd, bn_gamma, bn_beta))):
        raise _lib.Tnv3Error("conv3x3_wino43_dgrad_bnstats: unsupported shape, filter panel, z or per-channel constants mismatch")
    da = torch.empty((n, int(cout), h, w), dtype=torch.float32, device=dz.device)
    stats = torch.empty((int(cout), tiles, 2), dtype=torch.float64, device=dz.device)
    _lib.check(lib.tnv3_conv3x3_wino43_dgrad_bnstats(_lib.ptr(dz), _lib.ptr(u_t), _lib.ptr(da), _lib.ptr(stats), _lib.ptr(bn_z), _lib.ptr(bn_mean),
                                                     _lib.ptr(bn_invstd), _lib.ptr(bn_gamma), _lib.ptr(bn_beta), n, cin, int(cout), h, w, int(variant),
                                                     _lib.stream_ptr(dz)))
    return da, stats


def bn_relu_backward_tiles(da, z, gamma, beta, mean, invstd, tile_stats, inplace=True, out=None):
    """bn_relu_backward with the two per-channel sums already taken per pixel tile by conv3x3_wino_dgrad_bnstats: one pass over (dA, z).
    out: (dgamma, dbeta) destinations."""
    lib = _lib.load()
    _f32(da, z, gamma, beta, mean, invstd)
    _lib.dev_check(da, z, gamma, beta, mean, invstd, tile_stats)
    if tile_stats.dtype != torch.float64 or not tile_stats.is_contiguous():
        raise _lib.Tnv3Error("bn_relu_backward_tiles: tile_stats must be a contiguous float64 tensor")
    n, c, h, w = (int(v) for v in z.shape)
    dz = da if inplace else torch.empty_like(da)
    dgamma = _grad_out((c,), z.device, out[0] if out else None, "bn_relu_backward_tiles")
    dbeta = _grad_out((c,), z.device, out[1] if out else None, "bn_relu_backward_tiles")
    ws = _workspace(lib.tnv3_bn_workspace_bytes(c), z.device)
    _lib.check(lib.tnv3_bn_relu_backward_tiles(_lib.ptr(da), _lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd),
                                               _lib.ptr(tile_stats), int(tile_stats.shape[1]), _lib.ptr(dz), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                               _lib.ptr(ws), ws.numel() * 8, n, c, h * w, _lib.stream_ptr(z)))
    return dz, dgamma, dbeta


def conv3x3_dgrad(dz, wpack_t, c0, c1=0, cfg=-1):
    """dX = conv3x3(dZ, W^T flipped); returns (dx0 [N,c0,H,W], dx1 [N,c1,H,W] or None)."""
    lib = _lib.load()
    _f32(dz, wpack_t)
    _lib.dev_check(dz, wpack_t)
    n, cout, h, w = (int(v) for v in dz.shape)
    if wpack_t.numel() != lib.tnv3_conv3x3_packed_floats(cout, c0 + c1, 1):
        raise _lib.Tnv3Error("conv3x3_dgrad: packed (transposed) filter has the wrong size")
    dx0 = torch.empty((n, c0, h, w), dtype=torch.float32, device=dz.device)
    dx1 = torch.empty((n, c1, h, w), dtype=torch.float32, device=dz.device) if c1 else None
    _lib.check(lib.tnv3_conv3x3_dgrad(_lib.ptr(dz), _lib.ptr(wpack_t), _lib.ptr(dx0), _lib.ptr(dx1), n, cout, c0, c1, h, w,
                                      int(cfg), _lib.stream_ptr(dz)))
    return dx0, dx1


def conv3x3_wgrad(src0, dz, src1=None, up0=False, variant=None, out=None):
    """dW[Cout][C0+C1][3][3] for X = cat([up2x?(src0), src1], 1).  variant: kernel family for THIS call (None:
    tuning.WGRAD_VARIANT; 0 = the register-staged kernel -- the LDS-DMA staged twin 1 is in the diagnostics library).  out: where dW is written
    (see _grad_out)."""
    lib = _lib.load()
    _f32(src0, src1, dz)
    _lib.dev_check(src0, src1, dz)
    n, cout, h, w = (int(v) for v in dz.shape)
    c0 = int(src0.shape[1])
    c1 = int(src1.shape[1]) if src1 is not None else 0
    dw = _grad_out((cout, c0 + c1, 3, 3), dz.device, out, "conv3x3_wgrad")
    if variant is None:
        from . import tuning
        variant = tuning.WGRAD_VARIANT
    variant = max(0, int(variant))
    ws = _workspace(lib.tnv3_conv3x3_wgrad_workspace_bytes(n, c0, c1, cout, h, w, variant), dz.device)
    _lib.check(lib.tnv3_conv3x3_wgrad(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(ws), ws.numel() * 8,
                                      n, c0, c1, cout, h, w, int(bool(up0)), variant, _lib.stream_ptr(dz)))
    return dw


def wgrad_wino_supported(cin, cout, h, w):
    return bool(_lib.load().tnv3_conv3x3_wgrad_wino_supported(int(cin), int(cout), int(h), int(w)))


def _wgrad_wino_variant(variant, h=0):
    if variant is None:
        from . import tuning
        variant = tuning.WGRAD_WINO_VARIANT
        if int(variant) == 8 and h % 4:      # the F(4x4) kernel walks 4-row strips; other heights take the library's F(2x2) default
            variant = -1
    return int(variant)


def conv3x3_wgrad_wino(x, dz, variant=None, out=None):
    """dW[Cout][Cin][3][3] of a plain layer in Winograd form -- see tnv3_conv3x3_wgrad_wino.  variant: kernel for THIS
    call (None: tuning.WGRAD_WINO_VARIANT; -1 the library's pick: 8, the F(4x4, 3x3) kernel (H % 4 == 0, any Cin), else the F(2x2) kernels
    1 (role-split) or, when Cin % 64 != 0, 5 (every wave streams and transforms) -- 1 and 5 bit-identical.  0 / 2 / 3 / 4 / 6 / 7: twins of
    the diagnostics library)."""
    lib = _lib.load()
    _f32(x, dz)
    _lib.dev_check(x, dz)
    n, cout, h, w = (int(v) for v in dz.shape)
    cin = int(x.shape[1])
    if tuple(x.shape) != (n, cin, h, w):
        raise _lib.Tnv3Error("conv3x3_wgrad_wino: x and dz must share batch and spatial size")
    dw = _grad_out((cout, cin, 3, 3), dz.device, out, "conv3x3_wgrad_wino")
    ws = _workspace(lib.tnv3_conv3x3_wgrad_wino_workspace_bytes(n, cin, cout, h, w), dz.device)
    _lib.check(lib.tnv3_conv3x3_wgrad_wino(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(ws), ws.numel() * 8, n, cin, cout, h, w,
                                           _wgrad_wino_variant(variant, h), _lib.stream_ptr(dz)))
    return dw


_WGRAD_UP2X_OF_PLAIN = {-1: -1, 1: 2, 2: 2, 5: 5, 8: 8}      # tuning.WGRAD_WINO_VARIANT -> tnv3_conv3x3_wgrad_up2x's own numbering (2: the diag twin of 1)


def conv3x3_wgrad_up2x(x_low, skip, dz, wino_variant=None, out=None, up_variant=None):
    """dW[Cout][C0+C1][3][3] of a decoder-entry layer (X = cat([upsample2x(x_low), skip], 1)), its upsampled channels at the
    low resolution -- see tnv3_conv3x3_wgrad_up2x.  wino_variant: an explicit number is the C entry's own (-1 / 8: the library's default
    for the skip half -- F(4x4) where it applies -- with the Winograd form of the upsampled half; 2 / 5: F(2x2) kernel 1 / 5 for the skip half;
    1: kernel 1 and the upsampled half by the four 2x2-window launches); None follows tuning.WGRAD_WINO_VARIANT, the PLAIN layers' kernel
    choice (dispatchable: -1, 1, 2, 5, 8), keeping the Winograd form of the upsampled half: the plain kernels 1 and 2 both map to 2 here (the
    skip half has no kernel-2 form), 5 and 8 to themselves.  up_variant: the form of the upsampled half (None: tuning.WGRAD_UP2X_VARIANT;
    -1 the fastest the shape allows, 2 the 25-of-36 F(4x4) form, 1 the 9-GEMM F(2x2) form, 0 four 2x2-window launches)."""
    from . import tuning
    if wino_variant is None:
        wino_variant = _WGRAD_UP2X_OF_PLAIN[int(tuning.WGRAD_WINO_VARIANT)]
    if up_variant is None:
        up_variant = tuning.WGRAD_UP2X_VARIANT
    lib = _lib.load()
    _f32(x_low, skip, dz)
    _lib.dev_check(x_low, skip, dz)
    n, cout, h, w = (int(v) for v in dz.shape)
    c0, c1 = int(x_low.shape[1]), int(skip.shape[1])
    if tuple(x_low.shape[2:]) != (h // 2, w // 2) or tuple(skip.shape[2:]) != (h, w):
        raise _lib.Tnv3Error("conv3x3_wgrad_up2x: x_low must be half the size of skip / dz")
    dw = _grad_out((cout, c0 + c1, 3, 3), dz.device, out, "conv3x3_wgrad_up2x")
    ws = _workspace(lib.tnv3_conv3x3_wgrad_up2x_workspace_bytes(n, c0, c1, cout, h // 2, w // 2), dz.device)
    _lib.check(lib.tnv3_conv3x3_wgrad_up2x(_lib.ptr(x_low), _lib.ptr(skip), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(ws), ws.numel() * 8,
                                           n, c0, c1, cout, h // 2, w // 2, int(wino_variant), int(up_variant), _lib.stream_ptr(dz)))
    return dw


def wbce_forward(p, y, reduce=True):
    lib = _lib.load()
    _f32(p, y)
    _lib.dev_check(p, y)
    n = int(p.shape[0])
    per = p.numel() // n
    out = torch.empty(1 if reduce else n, dtype=torch.float32, device=p.device)
    ws = _workspace(lib.tnv3_wbce_workspace_bytes(n), p.device)
    _lib.check(lib.tnv3_wbce_forward(_lib.ptr(p), _lib.ptr(y), _lib.ptr(out), _lib.ptr(ws), ws.numel() * 8, n, per,
                                     int(bool(reduce)), _lib.stream_ptr(p)))
    return out


def wbce_backward(p, y, upstream, reduce=True):
    lib = _lib.load()
    _f32(p, y, upstream)
    _lib.dev_check(p, y, upstream)
    n = int(p.shape[0])
    per = p.numel() // n
    dp = torch.empty_like(p)
    _lib.check(lib.tnv3_wbce_backward(_lib.ptr(p), _lib.ptr(y), _lib.ptr(upstream), _lib.ptr(dp), n, per, int(bool(reduce)),
                                      _lib.stream_ptr(p)))
    return dp


def head_backward(dp, p, a, weight, out=None):
    """Backward of p = sigmoid(conv1x1(a) + b): returns (da, dW (L,64,1,1), db (L,)).  out: (dW, db) destinations."""
    lib = _lib.load()
    _f32(dp, p, a, weight)
    _lib.dev_check(dp, p, a, weight)
    n, l, h, w = (int(v) for v in p.shape)
    if int(a.shape[1]) != 64:
        raise _lib.Tnv3Error("head_backward: expects 64 input channels")
    da = torch.empty_like(a)
    dw = _grad_out((l, 64, 1, 1), p.device, out[0] if out else None, "head backward")
    db = _grad_out((l,), p.device, out[1] if out else None, "head backward")
    ws = _workspace(lib.tnv3_head_backward_workspace_bytes(l), p.device)
    _lib.check(lib.tnv3_head_backward(_lib.ptr(dp), _lib.ptr(p), _lib.ptr(a), _lib.ptr(weight), _lib.ptr(da), _lib.ptr(dw),
                                      _lib.ptr(db), _lib.ptr(ws), ws.numel() * 8, n, l, h * w, _lib.stream_ptr(p)))
    return da, dw, db


def head1x1_sigmoid_wbce(x, weight, bias, y, reduce=True):
    """(p, loss): the head (1x1 conv + bias + sigmoid) and WBCELoss(p, y, reduce) in ONE pass (tnv3_head1x1_sigmoid_wbce)."""
    lib = _lib.load()
    _f32(x, weight, bias, y)
    _lib.dev_check(x, weight, bias, y)
    n, c, h, w = (int(v) for v in x.shape)
    l = int(weight.shape[0])
    if tuple(y.shape) != (n, l, h, w):
        raise _lib.Tnv3Error("head1x1_sigmoid_wbce: y must have the heat maps' shape")
    p = torch.empty((n, l, h, w), dtype=torch.float32, device=x.device)
    loss = torch.empty(1 if reduce else n, dtype=torch.float32, device=x.device)
    ws = _workspace(lib.tnv3_head_wbce_workspace_bytes(n), x.device)
    _lib.check(lib.tnv3_head1x1_sigmoid_wbce(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(p), _lib.ptr(loss),
                                             _lib.ptr(ws), ws.numel() * 8, n, c, l, h * w, int(bool(reduce)), _lib.stream_ptr(x)))
    return p, loss


def head_wbce_backward(y, p, a, weight, upstream, reduce=True, out=None):
    """Backward of head + sigmoid + WBCELoss without a dP tensor: returns (da, dW (L,64,1,1), db (L,)).  out: (dW, db) destinations."""
    lib = _lib.load()
    _f32(y, p, a, weight, upstream)
    _lib.dev_check(y, p, a, weight, upstream)
    n, l, h, w = (int(v) for v in p.shape)
    if int(a.shape[1]) != 64:
        raise _lib.Tnv3Error("head_wbce_backward: expects 64 input channels")
    da = torch.empty_like(a)
    dw = _grad_out((l, 64, 1, 1), p.device, out[0] if out else None, "head backward")
    db = _grad_out((l,), p.device, out[1] if out else None, "head backward")
    ws = _workspace(lib.tnv3_head_backward_workspace_bytes(l), p.device)
    _lib.check(lib.tnv3_head_wbce_backward(_lib.ptr(y), _lib.ptr(p), _lib.ptr(a), _lib.ptr(weight), _lib.ptr(upstream), _lib.ptr(da),
                                           _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), ws.numel() * 8, n, l, h * w, int(bool(reduce)),
                                           _lib.stream_ptr(p)))
    return da, dw, db


def maxpool2x2_backward_add(x, dpool, dskip=None):
    lib = _lib.load()
    _f32(x, dpool, dskip)
    _lib.dev_check(x, dpool, dskip)
    n, c, h, w = (int(v) for v in x.shape)
    dx = torch.empty_like(x)
    _lib.check(lib.tnv3_maxpool2x2_backward_add(_lib.ptr(x), _lib.ptr(dpool), _lib.ptr(dskip), _lib.ptr(dx), n * c, h, w,
                                                _lib.stream_ptr(x)))
    return dx


def maxpool2x2_bnstats_supported(n, h, w):
    return _lib.load().tnv3_maxpool2x2_bn_stats_tiles(int(n), int(h), int(w)) > 0


def maxpool2x2_backward_add_bnstats(z, dpool, dskip, mean, invstd, gamma, beta):
    """maxpool2x2_backward_add for a pooled tensor that is ReLU(BatchNorm(z)) of a block normalised in this step: returns (dx, tile_stats) --
    tile_stats [C][slices][2] float64, that block's two BatchNorm-backward sums of dx, for bn_relu_backward_tiles (autograd of model.py:9-10
    behind model.py:48,51,54)."""
    lib = _lib.load()
    _f32(z, dpool, dskip, mean, invstd, gamma, beta)
    _lib.dev_check(z, dpool, dskip, mean, invstd, gamma, beta)
    n, c, h, w = (int(v) for v in z.shape)
    if tuple(dpool.shape) != (n, c, h // 2, w // 2) or (dskip is not None and dskip.shape != z.shape):
        raise _lib.Tnv3Error("maxpool2x2_backward_add_bnstats: shape mismatch")
    if not (z.is_contiguous() and dpool.is_contiguous() and (dskip is None or dskip.is_contiguous())):
        raise _lib.Tnv3Error("maxpool2x2_backward_add_bnstats: contiguous tensors required")
    slices = lib.tnv3_maxpool2x2_bn_stats_tiles(n, h, w)
    if slices <= 0:
        raise _lib.Tnv3Error(f"maxpool2x2_backward_add_bnstats: needs H % 2 == 0 and W % 4 == 0 (got {h}x{w})")
    dx = torch.empty_like(z)
    st = torch.empty((c, slices, 2), dtype=torch.float64, device=z.device)
    _lib.check(lib.tnv3_maxpool2x2_backward_add_bnstats(_lib.ptr(z), _lib.ptr(dpool), _lib.ptr(dskip), _lib.ptr(dx), _lib.ptr(mean), _lib.ptr(invstd),
                                                        _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(st), n, c, h, w, _lib.stream_ptr(z)))
    return dx, st


def upsample2x_backward(d_hi):
    lib = _lib.load()
    _f32(d_hi)
    _lib.dev_check(d_hi)
    n, c, h, w = (int(v) for v in d_hi.shape)
    d_lo = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=d_hi.device)
    _lib.check(lib.tnv3_upsample2x_backward(_lib.ptr(d_hi), _lib.ptr(d_lo), n * c, h // 2, w // 2, _lib.stream_ptr(d_hi)))
    return d_lo


def mixup(x, lam, perm):
    """out[n] = x[n]*lam[n] + x[perm[n]]*(1-lam[n]); lam float32 (N,), perm int32 (N,)."""
    lib = _lib.load()
    _f32(x, lam)
    _lib.dev_check(x, lam, perm)
    if perm.dtype != torch.int32:
        raise _lib.Tnv3Error("mixup: perm must be int32")
    n = int(x.shape[0])
    out = torch.empty_like(x)
    _lib.check(lib.tnv3_mixup(_lib.ptr(x), _lib.ptr(lam), _lib.ptr(perm), _lib.ptr(out), n, x.numel() // n, _lib.stream_ptr(x)))
    return out


# ------------------------------------------------------------------------------------------------- optimiser / mixup draws
def _ptr_array(tensors):
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = t.data_ptr() if t is not None else None
    return arr


def _numel_array(tensors):
    import ctypes
    return (ctypes.c_long * len(tensors))(*[int(t.numel()) for t in tensors])


def _check_tensor_lists(*lists):
    flat = [t for lst in lists if lst is not None for t in lst]
    _f32(*flat)
    _lib.dev_check(*flat)
    n = len(lists[0])
    for lst in lists:
        if lst is not None and len(lst) != n:
            raise _lib.Tnv3Error("tensor lists must have the same length")
    for lst in lists[1:]:
        if lst is not None and any(a.numel() != b.numel() for a, b in zip(lists[0], lst)):
            raise _lib.Tnv3Error("tensor lists must have matching sizes")
    return n


def grad_norm(grads, max_norm):
    """clip_grad_norm_'s two scalars on the device: returns a (2,) tensor [total L2 norm, min(1, max_norm / (norm + 1e-6))]."""
    lib = _lib.load()
    n = _check_tensor_lists(grads)
    out = torch.empty(2, dtype=torch.float32, device=grads[0].device)
    ws = _workspace(lib.tnv3_grad_norm_workspace_bytes(n), grads[0].device)
    _lib.check(lib.tnv3_grad_norm(_ptr_array(grads), _numel_array(grads), n, float(max_norm), _lib.ptr(out), _lib.ptr(ws), ws.numel() * 8,
                                  _lib.stream_ptr(grads[0])))
    return out


def adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, clip_coef=None,
              zero_grad=False):
    """torch.optim.Adam's update of every listed tensor in one launch (tnv3_adam_step); `step` is the 1-based update count;
    clip_coef: (1,) device tensor scaling the gradients first (grad_norm(...)[1:]), or None."""
    lib = _lib.load()
    n = _check_tensor_lists(params, grads, exp_avgs, exp_avg_sqs)
    if clip_coef is not None:
        _f32(clip_coef)
        _lib.dev_check(params[0], clip_coef)
    _lib.check(lib.tnv3_adam_step(_ptr_array(params), _ptr_array(grads), _ptr_array(exp_avgs), _ptr_array(exp_avg_sqs), _numel_array(params),
                                  n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                  _lib.ptr(clip_coef), int(bool(zero_grad)), _lib.stream_ptr(params[0])))


def sgd_step(params, grads, momentum_bufs, lr, momentum=0.0, weight_decay=0.0, first_step=False, clip_coef=None, zero_grad=False):
    """torch.optim.SGD's update (momentum, dampening 0, no Nesterov) of every listed tensor in one launch (tnv3_sgd_step)."""
    lib = _lib.load()
    n = _check_tensor_lists(params, grads, momentum_bufs if momentum else None)
    _lib.check(lib.tnv3_sgd_step(_ptr_array(params), _ptr_array(grads), _ptr_array(momentum_bufs) if momentum else None, _numel_array(params), n,
                                 float(lr), float(momentum), float(weight_decay), int(bool(first_step)), _lib.ptr(clip_coef),
                                 int(bool(zero_grad)), _lib.stream_ptr(params[0])))


def mixup_draw(n, alpha, seed, step, device):
    """(lam float32 (n,), perm int32 (n,)) of train.py:33-36 drawn ON THE DEVICE from Philox(seed, step): no host RNG, no H2D copy."""
    lib = _lib.load()
    device = torch.device(device)
    lam = torch.empty(int(n), dtype=torch.float32, device=device)
    perm = torch.empty(int(n), dtype=torch.int32, device=device)
    _lib.dev_check(lam, perm)
    with torch.cuda.device(device) if device.type == "cuda" else _nullcontext():
        _lib.check(lib.tnv3_mixup_draw(_lib.ptr(lam), _lib.ptr(perm), int(n), float(alpha), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1),
                                       _lib.stream_ptr(lam)))
    return lam, perm


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------------------------------------- InpaintNet backward
def conv1d_act_backward(dout, out, act, nlc=False):
    """dPre[N][C][L] = dOut * act'(out); nlc: dOut/out are (N, L, C)."""
    lib = _lib.load()
    _f32(dout, out)
    _lib.dev_check(dout, out)
    if nlc:
        n, l, c = (int(v) for v in out.shape)
    else:
        n, c, l = (int(v) for v in out.shape)
    dpre = torch.empty((n, c, l), dtype=torch.float32, device=out.device)
    _lib.check(lib.tnv3_conv1d_act_backward(_lib.ptr(dout), _lib.ptr(out), _lib.ptr(dpre), n, c, l, int(act), int(bool(nlc)),
                                            _lib.stream_ptr(out)))
    return dpre


def conv1d_k3_dgrad(dpre, weight, c0, c1=0, dx0=None, dx1=None):
    """dX = conv1d(dPre, W^T flipped) split into the two concat operands; pass existing dx0/dx1 to ACCUMULATE into them."""
    lib = _lib.load()
    _f32(dpre, weight, dx0, dx1)
    _lib.dev_check(dpre, weight, dx0, dx1)
    n, cout, l = (int(v) for v in dpre.shape)
    if tuple(weight.shape) != (cout, c0 + c1, 3):
        raise _lib.Tnv3Error("conv1d_k3_dgrad: weight shape mismatch")
    acc = (1 if dx0 is not None else 0) | (2 if (dx1 is not None and c1) else 0)
    if dx0 is None:
        dx0 = torch.empty((n, c0, l), dtype=torch.float32, device=dpre.device)
    if dx1 is None and c1:
        dx1 = torch.empty((n, c1, l), dtype=torch.float32, device=dpre.device)
    _lib.check(lib.tnv3_conv1d_k3_dgrad(_lib.ptr(dpre), _lib.ptr(weight), _lib.ptr(dx0), _lib.ptr(dx1 if c1 else None), n, cout,
                                        c0, c1, l, acc, _lib.stream_ptr(dpre)))
    return dx0, (dx1 if c1 else None)


def conv1d_k3_wgrad(src0, dpre, src1=None, src_nlc=False):
    """(dW [Cout][C0+C1][3], db [Cout]) for X = cat([src0, src1])."""
    lib = _lib.load()
    _f32(src0, src1, dpre)
    _lib.dev_check(src0, src1, dpre)
    n, cout, l = (int(v) for v in dpre.shape)
    c0 = int(src0.shape[2] if src_nlc else src0.shape[1])
    c1 = 0 if src1 is None else int(src1.shape[2] if src_nlc else src1.shape[1])
    dw = torch.empty((cout, c0 + c1, 3), dtype=torch.float32, device=dpre.device)
    db = torch.empty(cout, dtype=torch.float32, device=dpre.device)
    ws = _workspace(lib.tnv3_conv1d_k3_wgrad_workspace_bytes(n, c0, c1, cout), dpre.device)
    _lib.check(lib.tnv3_conv1d_k3_wgrad(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(dpre), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws),
                                        ws.numel() * 8, n, c0, c1, cout, l, int(bool(src_nlc)), _lib.stream_ptr(dpre)))
    return dw, db


# ------------------------------------------------------------------------------------------------- device guard
# The C ABI runs a call on its stream's device, but a tensor's *default* stream is the NULL stream (= "the calling thread's
# current device"), and the stream-less workspace queries plan for the current device too.  Every op therefore runs with the
# device of its first GPU tensor current (a no-op check when it already is, i.e. always in single-device processes).
_TENSOR_OPS = ["pack_conv3x3_weights", "bn_eval_scale", "pack_wino_weights", "pack_wino_weights_multi", "pack_wino43_weights", "conv3x3_wino43", "conv3x3_wino43_stats", "conv3x3_wino43_dgrad_bnstats", "conv3x3_wino", "conv3x3_wino_stats", "pack_up2x_weights", "conv_up2x", "pack_up2x_wino_weights", "conv_up2x_wino", "pack_dgrad_up2x_wino_weights", "dgrad_up2x_wino",
               "pack_dgrad_up2x_weights", "dgrad_up2x", "conv3x3", "head1x1_sigmoid", "maxpool2x2", "conv1d_k3", "inpaintnet_fused", "ensemble_frames",
               "heatmap_peakfind", "heatmap_box_max", "bn_train_forward", "bn_relu_backward", "bn_bwd_consts", "conv3x3_wino_dgrad_bnstats", "bn_relu_backward_tiles", "conv3x3_dgrad", "conv3x3_wgrad",
               "conv3x3_wgrad_wino", "conv3x3_wgrad_up2x", "wbce_forward", "wbce_backward", "head_backward", "head1x1_sigmoid_wbce", "head_wbce_backward",
               "maxpool2x2_backward_add", "maxpool2x2_backward_add_bnstats", "dgrad_up2x_wino_bnstats", "maxpool2x2_bnstats_supported", "upsample2x_backward", "mixup", "conv1d_act_backward", "conv1d_k3_dgrad", "conv1d_k3_wgrad",
               "grad_norm", "adam_step", "sgd_step", "inpaintnet_pack", "inpaintnet_pack_t", "inpaintnet_fused_train_forward", "inpaintnet_fused_backward"]          # list-of-tensor ops: the guard looks inside the lists
for _name in _TENSOR_OPS:
    globals()[_name] = _lib.on_tensor_device(globals()[_name])
del _name
