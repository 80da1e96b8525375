"""OpenCV pin for predict_location (test.py:52-79) -- the one call of the hot path whose arithmetic lives in a third-party
library that is NOT installable in the build container (opencv_python 4.4.0.46, requirements.txt:3), so the oracle's
restatement (8-connectivity, outer borders, boundingRect, contour list in reverse discovery order => equal-area ties go to
the component found LAST) is "parity unpinned" there.  These tests run wherever `cv2` imports: the first box that has OpenCV
pins the oracle AND the HIP kernel (including the tie switch) against the real findContours / boundingRect."""
import numpy as np
import pytest

from oracle import postproc as opp


def _cv2_largest_box(img_u8):
    """What test.py:52-79 computes, through the real OpenCV calls: external contours -> bounding rectangles -> the first
    rectangle of maximal area in OpenCV's own contour order (strict '>')."""
    import cv2
    if img_u8.max() == 0:
        return (0, 0, 0, 0)
    found = cv2.findContours(img_u8.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
    contours = found[0] if len(found) == 2 else found[1]            # OpenCV 3 returns (image, contours, hierarchy)
    rects = [cv2.boundingRect(c) for c in contours]
    best = 0
    for i in range(1, len(rects)):
        if rects[i][2] * rects[i][3] > rects[best][2] * rects[best][3]:
            best = i
    return tuple(int(v) for v in rects[best])


def _maps():
    from test_emu_postproc import _blob_maps, _wide_maps
    rng = np.random.RandomState(5)
    out = [m for m in _blob_maps()] + [m for m in _wide_maps()]
    # tie-heavy maps: many equal-area boxes in different raster positions
    for k in range(6):
        t = np.zeros((48, 64), np.float32)
        for _ in range(12):
            y, x = rng.randint(0, 45), rng.randint(0, 61)
            t[y:y + 2, x:x + 3] = 1
        out.append(t)
    big = np.zeros((288, 512), np.float32)
    big[100:103, 200:204] = 0.9; big[30:33, 400:404] = 0.8; big[250:253, 10:14] = 0.7       # three equal boxes at 288x512
    out.append(big)
    return out


def test_oracle_predict_location_equals_opencv():
    pytest.importorskip("cv2")
    assert opp.TIE_LAST_WINS is True
    for k, m in enumerate(_maps()):
        img = opp.to_img(m > 0.5)
        assert tuple(int(v) for v in opp.predict_location(img)) == _cv2_largest_box(img), k


@pytest.mark.gpu
def test_hip_peakfind_equals_opencv(gpu_device):
    pytest.importorskip("cv2")
    import torch
    from tracknetv3_amd import ops
    groups = {}
    for m in _maps():
        groups.setdefault(m.shape, []).append(m)
    for shape, ms in groups.items():
        got = ops.heatmap_peakfind(torch.from_numpy(np.stack(ms)).to(gpu_device), 0.5, tie_last_wins=True).cpu().numpy()
        want = np.array([_cv2_largest_box(opp.to_img(m > 0.5)) for m in ms])
        assert np.array_equal(got, want), shape
