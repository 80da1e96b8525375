"""ctypes binding of libtnv3_hip.so (C ABI: include/tracknetv3_hip.h).

The product path has NO fallback: if the HIP library cannot be loaded, every op raises.  The only other
library this module will ever bind is the CPU SIMT emulator built by the test-suite, and only when a test
calls ``use_library()`` explicitly (it is never searched for or loaded implicitly).
"""
import ctypes
import os

import torch

from . import _build

_c = ctypes
_f32p = _c.c_void_p
_lib = None
_is_emulator = False
ABI_VERSION = 8


class Tnv3Error(RuntimeError):
    pass


def _declare(lib):
    def sig(name, restype, *argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    i, p, f, sz, lg = _c.c_int, _c.c_void_p, _c.c_float, _c.c_size_t, _c.c_long
    ip = _c.POINTER(_c.c_int)
    sig("tnv3_abi_version", i)
    sig("tnv3_last_error", _c.c_char_p)
    sig("tnv3_conv3x3_num_configs", i)
    sig("tnv3_conv3x3_config_info", i, i, ip, ip, ip, ip, ip, ip)
    sig("tnv3_conv3x3_packed_floats", sz, i, i, i)
    sig("tnv3_pack_conv3x3_weights", i, p, p, i, i, i, p)
    sig("tnv3_bn_eval_scale", i, p, p, f, p, i, p)
    sig("tnv3_conv3x3_forward", i, p, p, p, p, p, p, p, i, i, i, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_forward_add", i, p, p, p, p, p, p, p, p, i, i, i, i, i, i, i, i, i, p)
    sig("tnv3_conv_up2x_packed_floats", sz, i, i)
    sig("tnv3_pack_up2x_weights", i, p, p, i, i, i, p)
    sig("tnv3_conv_up2x_forward", i, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv_up2x_wino_supported", i, i, i, i, i, i)
    sig("tnv3_conv_up2x_wino_packed_floats", sz, i, i, i)
    sig("tnv3_conv_up2x_wino_pack", i, p, p, i, i, i, i, p)
    sig("tnv3_conv_up2x_wino_forward", i, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_dgrad_up2x_wino_supported", i, i, i, i, i, i)
    sig("tnv3_dgrad_up2x_wino_packed_floats", sz, i, i, i)
    sig("tnv3_dgrad_up2x_wino_pack", i, p, p, i, i, i, i, p)
    sig("tnv3_dgrad_up2x_wino", i, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_dgrad_up2x_wino_bnstats", i, p, p, p, p, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_dgrad_up2x_packed_floats", sz, i, i)
    sig("tnv3_pack_dgrad_up2x_weights", i, p, p, i, i, i, p)
    sig("tnv3_dgrad_up2x", i, p, p, p, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wgrad_up2x_workspace_bytes", sz, i, i, i, i, i, i)
    sig("tnv3_conv3x3_wgrad_up2x", i, p, p, p, p, p, sz, i, i, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino_packed_floats", sz, i, i)
    sig("tnv3_conv3x3_wino_supported", i, i, i, i, i)
    sig("tnv3_conv3x3_wino_layout", i, i)
    sig("tnv3_conv3x3_wino_has_stats", i, i)
    sig("tnv3_conv3x3_wino_pick", i, i, i)
    sig("tnv3_conv3x3_wino_pack", i, p, p, i, i, i, p)
    sig("tnv3_conv3x3_wino_pack_view", i, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino_pack_multi", i, p, i, p)
    sig("tnv3_conv3x3_wino43_supported", i, i, i, i, i)
    sig("tnv3_conv3x3_wino43_packed_floats", sz, i, i, i)
    sig("tnv3_conv3x3_wino43_pack", i, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino43_forward", i, p, p, p, p, p, p, p, p, i, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino43_stats_tiles", lg, i, i, i, i)
    sig("tnv3_conv3x3_wino43_forward_stats", i, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino43_dgrad_bnstats", i, p, p, p, p, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino_forward", i, p, p, p, p, p, p, p, i, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wino_stats_tiles", lg, i, i, i, i)
    sig("tnv3_conv3x3_wino_forward_stats", i, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_bn_train_forward_tiles", i, p, p, lg, p, p, p, p, f, f, p, p, p, p, sz, i, i, i, p)
    sig("tnv3_bn_train_forward_tiles_pool", i, p, p, lg, p, p, p, p, f, f, p, p, p, p, p, sz, i, i, i, i, p)
    sig("tnv3_conv3x3_wgrad_wino_supported", i, i, i, i, i)
    sig("tnv3_conv3x3_wgrad_wino_workspace_bytes", sz, i, i, i, i, i)
    sig("tnv3_conv3x3_wgrad_wino", i, p, p, p, p, sz, i, i, i, i, i, i, p)
    sig("tnv3_head1x1_sigmoid", i, p, p, p, p, i, i, i, i, i, p)
    sig("tnv3_maxpool2x2", i, p, p, lg, i, i, p)
    sig("tnv3_conv1d_k3_forward", i, p, p, p, p, p, i, i, i, i, i, i, i, i, p)
    sig("tnv3_inpaintnet_packed_floats", sz)
    sig("tnv3_inpaintnet_pack", i, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p), p, p)
    sig("tnv3_inpaintnet_fused_forward", i, p, p, p, p, i, i, p)
    sig("tnv3_inpaintnet_packed_t_floats", sz)
    sig("tnv3_inpaintnet_act_floats", sz, i)
    sig("tnv3_inpaintnet_dpre_floats", sz, i)
    sig("tnv3_inpaintnet_param_floats", sz)
    sig("tnv3_inpaintnet_pack_t", i, _c.POINTER(_c.c_void_p), p, p)
    sig("tnv3_inpaintnet_fused_forward_train", i, p, p, p, p, p, i, i, p)
    sig("tnv3_inpaintnet_fused_backward", i, p, p, p, p, p, p, p, p, p, i, i, p)
    sig("tnv3_ensemble_frames", i, p, i, lg, i, i, p, lg, i, lg, i, p, p)
    sig("tnv3_peakfind_workspace_bytes", sz, i, i, i)
    sig("tnv3_heatmap_peakfind", i, p, f, i, p, p, sz, i, i, i, p)
    sig("tnv3_heatmap_box_max", i, p, p, p, i, i, i, p)
    sig("tnv3_bn_workspace_bytes", sz, i)
    sig("tnv3_bn_train_forward", i, p, p, p, p, p, f, f, p, p, p, p, sz, i, i, i, p)
    sig("tnv3_bn_relu_backward", i, p, p, p, p, p, p, p, p, p, p, p, sz, i, i, i, p)
    sig("tnv3_bn_bwd_consts", i, p, p, p, p, p, i, p)
    sig("tnv3_conv3x3_wino_dgrad_bnstats", i, p, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_bn_relu_backward_tiles", i, p, p, p, p, p, p, p, lg, p, p, p, p, sz, i, i, i, p)
    sig("tnv3_conv3x3_dgrad", i, p, p, p, p, i, i, i, i, i, i, i, p)
    sig("tnv3_conv3x3_wgrad_workspace_bytes", sz, i, i, i, i, i, i, i)
    sig("tnv3_conv3x3_wgrad", i, p, p, p, p, p, sz, i, i, i, i, i, i, i, i, p)
    sig("tnv3_wbce_workspace_bytes", sz, i)
    sig("tnv3_wbce_forward", i, p, p, p, p, sz, i, lg, i, p)
    sig("tnv3_wbce_backward", i, p, p, p, p, i, lg, i, p)
    sig("tnv3_head_backward_workspace_bytes", sz, i)
    sig("tnv3_head_backward", i, p, p, p, p, p, p, p, p, sz, i, i, i, p)
    sig("tnv3_head_wbce_workspace_bytes", sz, i)
    sig("tnv3_head1x1_sigmoid_wbce", i, p, p, p, p, p, p, p, sz, i, i, i, i, i, p)
    sig("tnv3_head_wbce_backward", i, p, p, p, p, p, p, p, p, p, sz, i, i, i, i, p)
    sig("tnv3_maxpool2x2_backward_add", i, p, p, p, p, lg, i, i, p)
    sig("tnv3_maxpool2x2_bn_stats_tiles", i, i, i, i)
    sig("tnv3_maxpool2x2_backward_add_bnstats", i, p, p, p, p, p, p, p, p, p, i, i, i, i, p)
    sig("tnv3_upsample2x_backward", i, p, p, lg, i, i, p)
    sig("tnv3_mixup", i, p, p, p, p, i, lg, p)
    dbl, pp, lp = _c.c_double, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_long)
    sig("tnv3_grad_norm_workspace_bytes", sz, i)
    sig("tnv3_grad_norm", i, pp, lp, i, f, p, p, sz, p)
    sig("tnv3_adam_step", i, pp, pp, pp, pp, lp, i, dbl, dbl, dbl, dbl, dbl, lg, p, i, p)
    sig("tnv3_sgd_step", i, pp, pp, pp, lp, i, dbl, dbl, dbl, i, p, i, p)
    sig("tnv3_mixup_draw", i, p, p, i, f, _c.c_uint64, _c.c_uint64, p)
    sig("tnv3_resample_bicubic_u8", i, p, p, p, p, p, p, p, i, p, p, p, i, p, i, i, i, i, i, i, p)
    sig("tnv3_median_u8", i, p, p, p, i, lg, p)
    sig("tnv3_absdiff_sum_u8", i, p, p, p, i, lg, p)
    sig("tnv3_conv1d_act_backward", i, p, p, p, i, i, i, i, i, p)
    sig("tnv3_conv1d_k3_dgrad", i, p, p, p, p, i, i, i, i, i, i, p)
    sig("tnv3_conv1d_k3_wgrad_workspace_bytes", sz, i, i, i, i)
    sig("tnv3_conv1d_k3_wgrad", i, p, p, p, p, p, p, sz, i, i, i, i, i, i, p)


EXPORTS = ["tnv3_abi_version", "tnv3_last_error", "tnv3_conv3x3_num_configs", "tnv3_conv3x3_config_info",
           "tnv3_conv3x3_packed_floats", "tnv3_pack_conv3x3_weights", "tnv3_bn_eval_scale", "tnv3_conv3x3_forward",
           "tnv3_head1x1_sigmoid", "tnv3_maxpool2x2", "tnv3_conv1d_k3_forward", "tnv3_inpaintnet_packed_floats", "tnv3_inpaintnet_pack", "tnv3_inpaintnet_fused_forward",
           "tnv3_inpaintnet_packed_t_floats", "tnv3_inpaintnet_act_floats", "tnv3_inpaintnet_dpre_floats", "tnv3_inpaintnet_param_floats",
           "tnv3_inpaintnet_pack_t", "tnv3_inpaintnet_fused_forward_train", "tnv3_inpaintnet_fused_backward",
           "tnv3_ensemble_frames",
           "tnv3_peakfind_workspace_bytes", "tnv3_heatmap_peakfind", "tnv3_bn_workspace_bytes", "tnv3_bn_train_forward",
           "tnv3_bn_relu_backward", "tnv3_bn_bwd_consts", "tnv3_conv3x3_wino_dgrad_bnstats", "tnv3_bn_relu_backward_tiles", "tnv3_conv3x3_dgrad", "tnv3_conv3x3_wgrad_workspace_bytes", "tnv3_conv3x3_wgrad",
           "tnv3_wbce_workspace_bytes", "tnv3_wbce_forward", "tnv3_wbce_backward", "tnv3_head_backward_workspace_bytes",
           "tnv3_head_backward", "tnv3_head_wbce_workspace_bytes", "tnv3_head1x1_sigmoid_wbce", "tnv3_head_wbce_backward",
           "tnv3_maxpool2x2_backward_add", "tnv3_maxpool2x2_bn_stats_tiles", "tnv3_maxpool2x2_backward_add_bnstats", "tnv3_dgrad_up2x_wino_bnstats", "tnv3_upsample2x_backward", "tnv3_mixup",
           "tnv3_grad_norm_workspace_bytes", "tnv3_grad_norm", "tnv3_adam_step", "tnv3_sgd_step", "tnv3_mixup_draw",
           "tnv3_resample_bicubic_u8", "tnv3_median_u8", "tnv3_absdiff_sum_u8", "tnv3_conv1d_act_backward", "tnv3_conv1d_k3_dgrad", "tnv3_conv1d_k3_wgrad_workspace_bytes", "tnv3_conv1d_k3_wgrad",
           "tnv3_heatmap_box_max", "tnv3_conv3x3_forward_add", "tnv3_conv_up2x_packed_floats",
           "tnv3_pack_up2x_weights", "tnv3_conv_up2x_forward", "tnv3_conv_up2x_wino_supported", "tnv3_conv_up2x_wino_packed_floats",
           "tnv3_conv_up2x_wino_pack", "tnv3_conv_up2x_wino_forward", "tnv3_dgrad_up2x_wino_supported", "tnv3_dgrad_up2x_wino_packed_floats",
           "tnv3_dgrad_up2x_wino_pack", "tnv3_dgrad_up2x_wino", "tnv3_dgrad_up2x_packed_floats", "tnv3_pack_dgrad_up2x_weights",
           "tnv3_dgrad_up2x", "tnv3_conv3x3_wgrad_up2x_workspace_bytes", "tnv3_conv3x3_wgrad_up2x",
           "tnv3_conv3x3_wgrad_wino_supported", "tnv3_conv3x3_wgrad_wino_workspace_bytes", "tnv3_conv3x3_wgrad_wino",
           "tnv3_conv3x3_wino_stats_tiles", "tnv3_conv3x3_wino_forward_stats", "tnv3_bn_train_forward_tiles", "tnv3_bn_train_forward_tiles_pool",
           "tnv3_conv3x3_wino_packed_floats", "tnv3_conv3x3_wino_layout", "tnv3_conv3x3_wino_has_stats", "tnv3_conv3x3_wino_pick", "tnv3_conv3x3_wino_supported", "tnv3_conv3x3_wino_pack", "tnv3_conv3x3_wino_pack_view", "tnv3_conv3x3_wino_pack_multi", "tnv3_conv3x3_wino_forward",
           "tnv3_conv3x3_wino43_supported", "tnv3_conv3x3_wino43_packed_floats", "tnv3_conv3x3_wino43_pack", "tnv3_conv3x3_wino43_forward",
           "tnv3_conv3x3_wino43_stats_tiles", "tnv3_conv3x3_wino43_forward_stats", "tnv3_conv3x3_wino43_dgrad_bnstats"]


def library_path():
    return _build.LIB


def load():
    """Load (building first if sources are newer and hipcc exists) the HIP library.  Raises if impossible."""
    global _lib, _is_emulator
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise Tnv3Error(f"libtnv3_hip.so is missing and could not be built: {e}") from e
    try:
        lib = _c.CDLL(path)
    except OSError as e:
        raise Tnv3Error(f"cannot load {path}: {e}") from e
    _declare(lib)
    if lib.tnv3_abi_version() != ABI_VERSION:
        raise Tnv3Error("libtnv3_hip.so ABI version mismatch")
    _lib, _is_emulator = lib, False
    return lib


def use_library(path):
    """TEST HOOK: bind an explicitly given library (the CPU SIMT emulator of tests/emu)."""
    global _lib, _is_emulator
    lib = _c.CDLL(path)
    _declare(lib)
    _lib = lib
    _is_emulator = hasattr(lib, "tnv3_is_emulator")
    return lib


def reset_library():
    global _lib, _is_emulator
    _lib, _is_emulator = None, False


def is_emulator():
    return _is_emulator


def check(rc):
    if rc != 0:
        raise Tnv3Error(f"tnv3 error {rc}: {load().tnv3_last_error().decode()}")


def stream_ptr(t):
    if t.is_cuda:
        return _c.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return _c.c_void_p(0)


def dev_check(*tensors):
    """All tensors: fp32 (or stated), contiguous, on one device; device must be a GPU unless the emulator is bound."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_contiguous():
            raise Tnv3Error("tnv3 ops need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise Tnv3Error(f"tensors on different devices: {t.device} vs {dev}")
    if dev is not None and not dev.type == "cuda" and not _is_emulator:
        raise Tnv3Error("tnv3 ops run on the GPU only (tensor is on %s); there is no CPU fallback" % dev)
    return dev


def ptr(t):
    return _c.c_void_p(t.data_ptr()) if t is not None else _c.c_void_p(0)


def on_tensor_device(fn):
    """Decorator: run `fn` with the device of its first tensor argument current (see the note at the end of ops.py)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        a = first_tensor(args)
        if a is not None and a.is_cuda and a.device.index != torch.cuda.current_device():
            with torch.cuda.device(a.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def first_tensor(args):
    """The first tensor among the positional arguments, looking inside list / tuple arguments too (the multi-tensor ops --
    grad_norm, adam_step, sgd_step, inpaintnet_pack -- take lists of tensors, pack_wino_weights_multi a list of tuples)."""
    for a in args:
        if isinstance(a, torch.Tensor):
            return a
        if isinstance(a, (list, tuple)):
            for b in a:
                if isinstance(b, torch.Tensor):
                    return b
                if isinstance(b, (list, tuple)):             # pack_wino_weights_multi: a list of (weight, c_from, flip)
                    for c in b:
                        if isinstance(c, torch.Tensor):
                            return c
    return None
