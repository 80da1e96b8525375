"""Frame preprocessing in front of TrackNet on the device: what Shuttlecock_Trajectory_Dataset does on
CPU workers for `frame_arr` inputs (dataset.py:101-109, 427-461) -- temporal median background, Pillow-exact BICUBIC
resize to 288x512, HWC->CHW, `/255.` -- as three HIP kernels.  Source frames stay uint8 in HBM; the network input is
assembled from the resized fp32 frames by `pipeline.predict_video`.
"""
import math

import numpy as np
import torch

from . import _lib
from .utils.general import HEIGHT, WIDTH

_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the BICUBIC filter over the full
    source range, in float64 like the C code: (xmin[out], xcount[out], coeff[out][ksize]) int32.  Host-side table: it
    depends only on the two sizes."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xmin = np.zeros(out_size, dtype=np.int32)
    xcnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        ww = 0.0
        for x in range(hi - lo):
            w = _bicubic((x + lo - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :hi - lo] /= ww
        xmin[xx], xcnt[xx] = lo, hi - lo
    scaled = kk * (1 << _PRECISION_BITS)
    return xmin, xcnt, np.where(kk < 0, np.trunc(-0.5 + scaled), np.trunc(0.5 + scaled)).astype(np.int32)


_tables = {}


def _device_tables(h, w, oh, ow, device):
    key = (h, w, oh, ow, str(device))
    if key not in _tables:
        cx, cy = resample_coeffs(w, ow), resample_coeffs(h, oh)
        # the kernels multiply pixel x coefficient as 24-bit integers (full-rate v_mad_i32_i24): Pillow's normalised weights stay below 1.13 * 2^22
        # (measured over size pairs from 1 to 1920); a table beyond 2^23 would be a different filter, and a wrong result
        if max(int(np.abs(cx[2]).max()), int(np.abs(cy[2]).max())) >= (1 << 23):
            raise _lib.Tnv3Error("resize_frames: a resampling coefficient does not fit 24 bits")
        xs = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in cx]
        ys = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in cy]
        lut = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)).to(device)   # `/= 255.` then .float()
        _tables[key] = (xs, ys, lut)
    return _tables[key]


@_lib.on_tensor_device
def resize_frames(frames_u8, out_h=HEIGHT, out_w=WIDTH, want_f32=True, want_u8=False, out=None):
    """(F, H, W, C) uint8 -> fp32 (F, C, out_h, out_w) in [0, 1] (and/or uint8 (F, out_h, out_w, C)); bit-exact with
    `np.moveaxis(np.array(Image.fromarray(img).resize((out_w, out_h))), -1, 0) / 255.`  out: where the fp32 planes go (a contiguous
    (F, C, out_h, out_w) fp32 tensor, e.g. a frame range of the caller's stream buffer)."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
        raise _lib.Tnv3Error("resize_frames: expected a (F, H, W, C) uint8 tensor")
    _lib.dev_check(frames_u8)
    f, h, w, c = (int(v) for v in frames_u8.shape)
    dev = frames_u8.device
    (xmin, xcnt, kkx), (ymin, ycnt, kky), lut = _device_tables(h, w, out_h, out_w, dev)
    if out is not None and (not want_f32 or out.dtype != torch.float32 or tuple(out.shape) != (f, c, out_h, out_w) or not out.is_contiguous() or out.device != dev):
        raise _lib.Tnv3Error("resize_frames: `out` must be a contiguous fp32 (F, C, out_h, out_w) tensor on the frames' device")
    out_f = (out if out is not None else torch.empty((f, c, out_h, out_w), dtype=torch.float32, device=dev)) if want_f32 else None
    out_u = torch.empty((f, out_h, out_w, c), dtype=torch.uint8, device=dev) if want_u8 else None
    if f == 0:
        return (out_f, out_u) if (want_f32 and want_u8) else (out_f if want_f32 else out_u)
    tmp = torch.empty((f, h, out_w, c), dtype=torch.uint8, device=dev)
    _lib.check(lib.tnv3_resample_bicubic_u8(_lib.ptr(frames_u8), _lib.ptr(tmp), _lib.ptr(out_f), _lib.ptr(out_u), _lib.ptr(xmin),
                                            _lib.ptr(xcnt), _lib.ptr(kkx), int(kkx.shape[1]), _lib.ptr(ymin), _lib.ptr(ycnt),
                                            _lib.ptr(kky), int(kky.shape[1]), _lib.ptr(lut), f, h, w, c, out_h, out_w,
                                            _lib.stream_ptr(frames_u8)))
    return (out_f, out_u) if (want_f32 and want_u8) else (out_f if want_f32 else out_u)


@_lib.on_tensor_device
def median_background(frames_u8, doubled=False):
    """np.median(frame_arr, 0) of a (T, H, W, C) uint8 stack.  Default: `.astype('uint8')` -> (H, W, C) uint8 (bg_mode
    'concat').  doubled=True: twice the float median as int16-range uint16 (exact, halves included) for the
    difference-frame modes."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
        raise _lib.Tnv3Error("median_background: expected a (T, H, W, C) uint8 tensor")
    _lib.dev_check(frames_u8)
    t = int(frames_u8.shape[0])
    shape = tuple(frames_u8.shape[1:])
    med = torch.empty(shape, dtype=torch.int16 if doubled else torch.uint8, device=frames_u8.device)
    _lib.check(lib.tnv3_median_u8(_lib.ptr(frames_u8), None if doubled else _lib.ptr(med), _lib.ptr(med) if doubled else None, t,
                                  med.numel(), _lib.stream_ptr(frames_u8)))
    return med


@_lib.on_tensor_device
def difference_frames(frames_u8, median_x2):
    """(F, H, W, 3) uint8 and the doubled median (H, W, 3) -> (F, H, W, 1) uint8 = uint8(sum_c |frame - median|)."""
    lib = _lib.load()
    _lib.dev_check(frames_u8, median_x2)
    f, h, w, c = (int(v) for v in frames_u8.shape)
    if c != 3 or median_x2.dtype != torch.int16 or tuple(median_x2.shape) != (h, w, 3):
        raise _lib.Tnv3Error("difference_frames: expected RGB frames and an int16 (H, W, 3) doubled median")
    out = torch.empty((f, h, w, 1), dtype=torch.uint8, device=frames_u8.device)
    if f:
        _lib.check(lib.tnv3_absdiff_sum_u8(_lib.ptr(frames_u8), _lib.ptr(median_x2), _lib.ptr(out), f, h * w, _lib.stream_ptr(frames_u8)))
    return out


def preprocess_video(frames_u8, bg_mode="concat", median_u8=None, chunk=64):
    """Source-resolution uint8 frames (T, H, W, 3) on the device -> (per-frame fp32 planes (T, C, 288, 512), median fp32
    (3, 288, 512) or None) as the reference's dataset produces them (dataset.py:427-461):
    bg_mode '' / 'concat': C = 3 (RGB; 'concat' also returns the resized median image that goes first in every window);
    'subtract': C = 1 (difference frame); 'subtract_concat': C = 4 (RGB + difference frame)."""
    if bg_mode not in ("", None, "concat", "subtract", "subtract_concat"):
        raise ValueError(f"unknown bg_mode '{bg_mode}'")
    med, med2 = None, None
    if bg_mode == "concat":
        if median_u8 is None:
            median_u8 = median_background(frames_u8)
        med = resize_frames(median_u8.unsqueeze(0))[0]
    elif bg_mode in ("subtract", "subtract_concat"):
        med2 = median_background(frames_u8, doubled=True)
    outs = []
    for s in range(0, int(frames_u8.shape[0]), chunk):
        part = frames_u8[s:s + chunk]
        planes = []
        if bg_mode != "subtract":
            planes.append(resize_frames(part))
        if med2 is not None:
            planes.append(resize_frames(difference_frames(part, med2)))
        outs.append(planes[0] if len(planes) == 1 else torch.cat(planes, 1))
    return torch.cat(outs, 0), med


class LazyResizer:
    """preprocess_video for bg_mode '' / 'concat' with the resize done ON DEMAND, a chunk of frames at a time, on whichever stream asks first:
    `out` (T, 3, 288, 512) is allocated at once, `ensure(lo, hi)` makes frames lo .. hi - 1 of it valid for the CURRENT stream (resizing the
    chunks nobody has resized yet there, waiting for the events of those another stream did).  pipeline.predict_video calls it from the side
    stream of each TrackNet batch, so the HBM-bound resize of batch k + 1 runs beside the MFMA-bound network of batch k instead of in front of
    everything (round 6: 1.8 of the 23 ms of a 256-frame nonoverlap run).  The temporal median (which every window needs) is taken up front.
    Same kernels, same bits as preprocess_video."""

    def __init__(self, frames_u8, bg_mode="concat", median_u8=None, chunk=64):
        if bg_mode not in ("", None, "concat"):
            raise ValueError("LazyResizer serves bg_mode '' / 'concat' (the difference-frame modes go through preprocess_video)")
        self.src, self.chunk, self.t = frames_u8, int(chunk), int(frames_u8.shape[0])
        self.median = None
        if bg_mode == "concat":
            if median_u8 is None:
                median_u8 = median_background(frames_u8)
            self.median = resize_frames(median_u8.unsqueeze(0))[0]
        self.out = torch.empty((self.t, int(frames_u8.shape[3]), HEIGHT, WIDTH), dtype=torch.float32, device=frames_u8.device)
        self.done = {}                                       # chunk -> event recorded behind its resize

    def ensure(self, lo, hi):
        cur = torch.cuda.current_stream(self.src.device) if self.src.is_cuda else None
        for c in range(max(0, int(lo)) // self.chunk, (min(self.t, int(hi)) - 1) // self.chunk + 1):
            ev = self.done.get(c)
            if ev is None:
                a, b = c * self.chunk, min(self.t, (c + 1) * self.chunk)
                resize_frames(self.src[a:b], out=self.out[a:b])
                if cur is not None:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                self.done[c] = ev if ev is not None else True
            elif cur is not None and ev is not True:
                cur.wait_event(ev)
