"""CPU restatement of the frame preprocessing that feeds TrackNet (the step just BEFORE the path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference lines followed:
* median background over the frame array, float64 median then `.astype('uint8')`      dataset.py:101-109, 748-781
* `Image.fromarray(img).resize(size=(WIDTH, HEIGHT))` -- Pillow's default BICUBIC with antialias support scaling, 8-bit
  fixed-point two-pass resample (horizontal then vertical)                              dataset.py:447-451, 629
* `np.moveaxis(img, -1, 0)`, median image FIRST for bg_mode 'concat', `frames /= 255.` in float64, then the training /
  inference loop casts `.float()`                                                       dataset.py:452-459; train.py:86

The resample arithmetic lives in Pillow (un-vendored third-party dependency, requirements.txt pins Pillow==10.0.0; this
image has Pillow 12.x).  Its published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here and PINNED bit-exactly against the installed Pillow by
tests/test_oracle_golden.py (live) -- the resample code has been stable across those versions.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size, support=2.0):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for a full-range box: (xmin[out], xcount[out], kk_int[out][ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    ss = 1.0 / filterscale
    bounds_min = np.zeros(out_size, dtype=np.int32)
    bounds_cnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                kk[xx, x] /= ww
        bounds_min[xx], bounds_cnt[xx] = xmin, xmax
    kk_int = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int32)
    return bounds_min, bounds_cnt, kk_int


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img, out_w, out_h):
    """(H, W, C) uint8 -> (out_h, out_w, C) uint8, bit-exact with PIL.Image.resize((out_w, out_h)) (default BICUBIC)."""
    h, w, _ = img.shape
    src = img.astype(np.int64)
    if out_w != w:
        xmin, xcnt, kk = resample_coeffs(w, out_w)
        tmp = np.empty((h, out_w, img.shape[2]), dtype=np.uint8)
        for xx in range(out_w):
            k = kk[xx, :xcnt[xx]].astype(np.int64)
            acc = (src[:, xmin[xx]:xmin[xx] + xcnt[xx], :] * k[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    else:
        tmp = img
    if out_h != h:
        ymin, ycnt, kk = resample_coeffs(h, out_h)
        out = np.empty((out_h, out_w, img.shape[2]), dtype=np.uint8)
        for yy in range(out_h):
            k = kk[yy, :ycnt[yy]].astype(np.int64)
            acc = (src[ymin[yy]:ymin[yy] + ycnt[yy]] * k[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        return out
    return tmp.copy()


def median_u8(frame_arr):
    """np.median(frame_arr, 0).astype('uint8') (dataset.py:103-105): float64 median, truncation toward zero."""
    return np.median(frame_arr, 0).astype("uint8")


def normalise_u8(img_u8):
    """`frames /= 255.` in float64 followed by `.float()` (dataset.py:459; train.py:86 / predict.py:138)."""
    return (img_u8.astype(np.float64) / 255.0).astype(np.float32)


def diff_frame_u8(img_u8, median_f64):
    """`np.sum(np.absolute(img - median_img), 2).astype('uint8')` (dataset.py:439, 443): the median stays FLOAT64 here
    (np.median result, halves possible), the channel sum can reach 765 and the uint8 cast truncates then wraps mod 256."""
    return np.sum(np.absolute(img_u8 - median_f64), 2).astype("uint8")


def tracknet_input_from_frames(frame_arr, starts, seq_len, bg_mode, height=288, width=512, median=None):
    """frame_arr (T, H, W, 3) uint8 RGB -> float32 (B, C, height, width) exactly as Shuttlecock_Trajectory_Dataset
    .__getitem__ does for frame_arr inputs (dataset.py:427-461), all four bg_mode values."""
    if bg_mode not in ("", None, "concat", "subtract", "subtract_concat"):
        raise ValueError(bg_mode)
    med = None
    if bg_mode and median is None:
        median = np.median(frame_arr, 0)                     # float64 (dataset.py:103-104)
    if bg_mode == "concat":
        med = np.moveaxis(resize_bicubic_u8(median.astype("uint8"), width, height), -1, 0)
    out = []
    for s in starts:
        chans = [] if med is None else [med]
        for f in range(seq_len):
            img = frame_arr[s + f]
            if bg_mode == "subtract":
                chans.append(resize_bicubic_u8(diff_frame_u8(img, median)[..., None], width, height)[None, ..., 0])
            elif bg_mode == "subtract_concat":
                chans.append(np.moveaxis(resize_bicubic_u8(img, width, height), -1, 0))
                chans.append(resize_bicubic_u8(diff_frame_u8(img, median)[..., None], width, height)[None, ..., 0])
            else:
                chans.append(np.moveaxis(resize_bicubic_u8(img, width, height), -1, 0))
        out.append(normalise_u8(np.concatenate(chans, 0)))
    return np.stack(out, 0)
