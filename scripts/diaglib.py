"""ctypes binding of libtnv3_diag.so (include/tracknetv3_hip_diag.h) for the measurement scripts.  The product package never
loads this library: its timing twins write wrong results by design."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracknetv3_amd import _build, _lib  # noqa: E402

_diag = None


def load():
    global _diag
    if _diag is None:
        path = _build.DIAG_LIB if os.path.exists(_build.DIAG_LIB) else _build.build_diag()
        lib = ctypes.CDLL(path)
        i, p = ctypes.c_int, ctypes.c_void_p
        lib.tnv3_diag_last_error.restype = ctypes.c_char_p
        for name, args in (("tnv3_diag_mfma_f32_probe", [p, i, i, p]),
                           ("tnv3_diag_conv3x3_forward", [p, p, p, i, i, i, i, i, i, i, p]),
                           ("tnv3_diag_conv3x3_wino_forward", [p, p, p, i, i, i, i, i, i, p]),
                           ("tnv3_diag_conv3x3_wino43_timeline", [p, p, p, p, i, i, i, i, i, i, p]),
                           ("tnv3_diag_conv3x3_wino43s_timeline", [p, p, p, p, i, i, i, i, i, i, i, i, i, p]),
                           ("tnv3_diag_conv3x3_wgrad_wino", [p, p, p, p, ctypes.c_size_t, i, i, i, i, i, i, p]),
                           ("tnv3_diag_coissue_probe", [p, p, i, i, i, i, p])):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = i, args
        _diag = lib
    return _diag


def check(rc):
    if rc != 0:
        raise RuntimeError(f"tnv3_diag error {rc}: {load().tnv3_diag_last_error().decode()}")


def mfma_f32_probe(buf, blocks, iters):
    check(load().tnv3_diag_mfma_f32_probe(_lib.ptr(buf), int(blocks), int(iters), _lib.stream_ptr(buf)))


def conv3x3_forward(x, wpack, y, cfg, diag):
    n, cin, h, w = (int(v) for v in x.shape)
    check(load().tnv3_diag_conv3x3_forward(_lib.ptr(x), _lib.ptr(wpack), _lib.ptr(y), n, cin, int(y.shape[1]), h, w, int(cfg), int(diag),
                                           _lib.stream_ptr(x)))


def conv3x3_wino_forward(x, u, y, variant):
    n, cin, h, w = (int(v) for v in x.shape)
    check(load().tnv3_diag_conv3x3_wino_forward(_lib.ptr(x), _lib.ptr(u), _lib.ptr(y), n, cin, int(y.shape[1]), h, w, int(variant),
                                                _lib.stream_ptr(x)))


def conv3x3_wino43_timeline(x, u, y, tl, variant=1):
    """Plain F(4x4) forward (y correct for variant 1) + phase totals of one mid-grid workgroup in tl (int64 tensor of 64 elements:
    [wave 8][8]); variants 2-4: the write-out without its stores / its LDS exchange / both (wrong results)."""
    n, cin, h, w = (int(v) for v in x.shape)
    check(load().tnv3_diag_conv3x3_wino43_timeline(_lib.ptr(x), _lib.ptr(u), _lib.ptr(y), _lib.ptr(tl), n, cin, int(y.shape[1]), h, w,
                                                   int(variant), _lib.stream_ptr(x)))


def coissue_probe(out_u64, gsrc, blocks, role_a, role_b, iters):
    check(load().tnv3_diag_coissue_probe(_lib.ptr(out_u64), _lib.ptr(gsrc), int(blocks), int(role_a), int(role_b), int(iters),
                                         _lib.stream_ptr(gsrc)))


def conv3x3_wgrad_wino(x, dz, variant):
    """dW through the diag library (timing twins 101-103 of the third-generation Winograd weight-gradient kernel: WRONG results)."""
    import torch
    from tracknetv3_amd import ops
    n, cout, h, w = (int(v) for v in dz.shape)
    cin = int(x.shape[1])
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dz.device)
    ws = ops._workspace(_lib.load().tnv3_conv3x3_wgrad_wino_workspace_bytes(n, cin, cout, h, w), dz.device)
    check(load().tnv3_diag_conv3x3_wgrad_wino(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(ws), ws.numel() * 8, n, cin, cout, h, w,
                                              int(variant), _lib.stream_ptr(dz)))
    return dw


def conv3x3_wino43s_timeline(x, u, y, tl, cbw=4, grow=13, ts=12, mask=0):
    """The 16x16x4 F(4x4) forward (y correct for mask 0) + phase totals of one mid-grid workgroup in tl (int64 tensor of 64 elements:
    [wave 8][8] = prologue, steps, write-outs, steps walked, tiles walked); mask: timing twins (wrong results)."""
    n, cin, h, w = (int(v) for v in x.shape)
    check(load().tnv3_diag_conv3x3_wino43s_timeline(_lib.ptr(x), _lib.ptr(u), _lib.ptr(y), _lib.ptr(tl), n, cin, int(y.shape[1]), h, w,
                                                    int(cbw), int(grow), int(ts), int(mask), _lib.stream_ptr(x)))
