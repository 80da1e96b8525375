"""Tensor-level wrappers over the C ABI (include/tracknetv3_hip.h).  Plumbing only: argument checks, output
allocation from the PyTorch caching allocator, current-stream hand-off.  All arithmetic is in the HIP library."""
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


def bn_fold(gamma, beta, running_mean, running_var, eps=BN_EPS):
    lib = _lib.load()
    _f32(gamma, beta, running_mean, running_var)
    _lib.dev_check(gamma, beta, running_mean, running_var)
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    _lib.check(lib.tnv3_bn_fold(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean), _lib.ptr(running_var),
                                float(eps), _lib.ptr(scale), _lib.ptr(shift), c, _lib.stream_ptr(gamma)))
    return scale, shift


def conv3x3(src0, wpack, cout, src1=None, scale=None, shift=None, up0=False, relu=False, cfg=-1, out=None):
    """act(conv3x3(cat([up2x?(src0), src1], 1), W) * scale + shift) -- see tnv3_conv3x3_forward."""
    lib = _lib.load()
    _f32(src0, src1, wpack, scale, shift, out)
    _lib.dev_check(src0, src1, wpack, scale, shift, out)
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
    _lib.check(lib.tnv3_conv3x3_forward(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(wpack), _lib.ptr(scale), _lib.ptr(shift),
                                        _lib.ptr(out), n, c0, c1, cout, h, w, int(bool(up0)), int(bool(relu)), int(cfg),
                                        _lib.stream_ptr(src0)))
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
    _lib.check(lib.tnv3_conv1d_k3_forward(_lib.ptr(src0), _lib.ptr(src1), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out),
                                          n, c0, c1, cout, l, int(bool(src_nlc)), int(bool(dst_nlc)), int(act),
                                          _lib.stream_ptr(src0)))
    return out
