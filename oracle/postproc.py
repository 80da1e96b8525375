"""CPU restatement of the TrackNetV3 heatmap -> coordinate post-process.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference lines followed:
* get_ensemble_weight                     test.py:25-50
* predict_location                        test.py:52-79   (cv2.findContours + cv2.boundingRect)
* generate_inpaint_mask                   test.py:223-258
* to_img                                  utils/general.py:110-122
* predict                                 predict.py:14-69
* heat-map temporal ensemble loop         predict.py:163-209
* coordinate temporal ensemble loop       predict.py:243-301
* COOR_TH etc.                            utils/general.py:15-19

PARITY UNPINNED at one boundary: ``predict_location`` calls into OpenCV
(opencv_python==4.4.0.46, requirements.txt:3), which is not vendored in the
reference and not installed in this image, and the reference ships no tests or
golden vectors for it.  The restatement follows OpenCV's documented semantics
(8-connected foreground components, RETR_EXTERNAL outer borders only,
boundingRect = (min_x, min_y, max_x-min_x+1, max_y-min_y+1)); contour order is
"last discovered in raster scan first", which together with the strict ``>`` at
test.py:74 means: among equal-area boxes the component whose first raster
pixel comes LAST wins.  That tie rule is a single switch (``TIE_LAST_WINS``).
``scipy.ndimage.label`` is used as an independent cross-check of the
component / bounding-box part in tests/, and ``suzuki_abe_external`` below restates
the PUBLISHED algorithm behind findContours (Suzuki & Abe 1985, Algorithm 2: outermost
borders by border following) as an independent derivation of which contours exist, in
which order they are discovered and what their boxes are; tests assert that it selects
the same box as the flood-fill restatement on tie maps, nested blobs and random maps.
Only "OpenCV returns the list newest-first" remains an implementation fact that needs
OpenCV itself (tests/test_cv2_pin.py).
"""
import math

import numpy as np

HEIGHT = 288
WIDTH = 512
SIGMA = 2.5
DELTA_T = 1 / math.sqrt(HEIGHT ** 2 + WIDTH ** 2)
COOR_TH = DELTA_T * 50

TIE_LAST_WINS = True   # see module docstring


def get_ensemble_weight(seq_len, eval_mode):
    """test.py:39-50."""
    if eval_mode == "average":
        w = np.ones(seq_len, dtype=np.float32) / np.float32(seq_len)
    elif eval_mode == "weight":
        w = np.ones(seq_len, dtype=np.float32)
        for i in range(math.ceil(seq_len / 2)):
            w[i] = i + 1
            w[seq_len - i - 1] = i + 1
        w = w / w.sum(dtype=np.float32)
    else:
        raise ValueError("Invalid mode")
    return w.astype(np.float32)


def to_img(image):
    """utils/general.py:120-122."""
    return (image * 255).astype("uint8")


def connected_boxes(binary):
    """8-connected foreground components in raster discovery order.

    Returns [(x, y, w, h, first_pixel_linear_index)] -- what cv2.findContours(RETR_EXTERNAL)
    + cv2.boundingRect yield for outer borders.  (Components nested inside another
    component's hole are also listed here; their boxes are strictly inside the outer
    box so they can never win or tie on area -- SURVEY 8c.)
    """
    h, w = binary.shape
    fg = binary != 0
    seen = np.zeros((h, w), dtype=bool)
    boxes = []
    ys, xs = np.nonzero(fg)
    for y0, x0 in zip(ys.tolist(), xs.tolist()):
        if seen[y0, x0]:
            continue
        stack = [(y0, x0)]
        seen[y0, x0] = True
        x_min = x_max = x0
        y_min = y_max = y0
        while stack:
            y, x = stack.pop()
            x_min, x_max = min(x_min, x), max(x_max, x)
            y_min, y_max = min(y_min, y), max(y_max, y)
            for dy in (-1, 0, 1):
                yy = y + dy
                if yy < 0 or yy >= h:
                    continue
                for dx in (-1, 0, 1):
                    xx = x + dx
                    if xx < 0 or xx >= w or seen[yy, xx] or not fg[yy, xx]:
                        continue
                    seen[yy, xx] = True
                    stack.append((yy, xx))
        boxes.append((x_min, y_min, x_max - x_min + 1, y_max - y_min + 1, y0 * w + x0))
    return boxes


# ---- an INDEPENDENT derivation of what cv2.findContours(RETR_EXTERNAL) returns: Suzuki & Abe's border following ------------
# S. Suzuki, K. Abe, "Topological structural analysis of digitized binary images by border following", CVGIP 30 (1985),
# the algorithm OpenCV's findContours implements (its documentation cites it as [Suzuki85]).  Algorithm 2 of the paper follows
# only the OUTERMOST borders -- RETR_EXTERNAL -- and is restated here step by step with the paper's step numbers, for
# 8-connected 1-pixels (4-connected 0-pixels), on a frame padded with one ring of zeros (OpenCV >= 3.2 pads the same way).
# It shares no code and no idea with `connected_boxes` (flood fill): borders are traced pixel by pixel, components nested
# in another component's hole are really absent, and boxes come from the traced points.  What the PAPER fixes is the
# discovery order (raster order of each outer border's starting point); that OpenCV hands the list back newest-first is a
# property of its legacy contour tree (every new contour is linked in FRONT of its siblings) which a paper cannot supply --
# tests/test_cv2_pin.py pins that last step wherever OpenCV is installed.
_SA_CW = ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1))      # clockwise on the screen (rows grow downwards)


def suzuki_abe_external(binary):
    """Outermost borders of the non-zero pixels in discovery order: [(start_y, start_x, [(y, x), ...])] (unpadded coordinates)."""
    h, w = binary.shape
    f = np.zeros((h + 2, w + 2), dtype=np.int32)
    f[1:-1, 1:-1] = (np.asarray(binary) != 0)
    contours = []
    for i in range(1, h + 1):
        lnbd = 0                                                  # Algorithm 2: LNBD is reset at the start of every row
        for j in np.nonzero(f[i])[0].tolist():                    # marks stay non-zero, so the set of visited columns is fixed
            if f[i, j] == 1 and f[i, j - 1] == 0 and lnbd <= 0:   # (1a) outer-border start, followed only when LNBD <= 0
                pts = _sa_follow(f, i, j, i, j - 1)
                contours.append((i - 1, j - 1, [(y - 1, x - 1) for y, x in pts]))
            if f[i, j] != 1:                                      # (4) LNBD <- f_ij (signed: +2 entering, -2 leaving a component)
                lnbd = int(f[i, j])
    return contours


def _sa_follow(f, i, j, i2, j2):
    """Step (3) of the paper with the marks +2 / -2 of Algorithm 2; returns the border's pixels in tracing order."""
    # (3.1) clockwise from (i2, j2) around (i, j): the first non-zero pixel (i1, j1)
    d0 = _SA_CW.index((i2 - i, j2 - j))
    first = None
    for k in range(8):
        dy, dx = _SA_CW[(d0 + k) % 8]
        if f[i + dy, j + dx] != 0:
            first = (i + dy, j + dx)
            break
    if first is None:
        f[i, j] = -2                                              # an isolated pixel
        return [(i, j)]
    i1, j1 = first
    i2, j2 = i1, j1                                               # (3.2)
    i3, j3 = i, j
    pts = []
    while True:
        # (3.3) counter-clockwise around (i3, j3), starting from the element after (i2, j2): the first non-zero pixel (i4, j4)
        d = _SA_CW.index((i2 - i3, j2 - j3))
        east_zero_examined = False
        for k in range(1, 9):
            dy, dx = _SA_CW[(d - k) % 8]
            if f[i3 + dy, j3 + dx] != 0:
                i4, j4 = i3 + dy, j3 + dx
                break
            if (dy, dx) == (0, 1):
                east_zero_examined = True
        # (3.4)
        if east_zero_examined:
            f[i3, j3] = -2
        elif f[i3, j3] == 1:
            f[i3, j3] = 2
        pts.append((i3, j3))
        # (3.5)
        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
            return pts
        i2, j2 = i3, j3
        i3, j3 = i4, j4


def predict_location_suzuki(heatmap, newest_first=True):
    """test.py:52-79 with the contour list derived by border following: boxes from the traced points (cv2.boundingRect of a
    point set = (min x, min y, max x - min x + 1, max y - min y + 1)), list order = discovery order reversed when
    `newest_first` (OpenCV's legacy contour tree), then the reference's strict-`>` scan."""
    if np.amax(heatmap) == 0:
        return 0, 0, 0, 0
    rects = []
    for _, _, pts in suzuki_abe_external(heatmap):
        ys, xs = [p[0] for p in pts], [p[1] for p in pts]
        rects.append((min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1))
    if newest_first:
        rects = rects[::-1]
    max_area_idx, max_area = 0, rects[0][2] * rects[0][3]
    for k in range(1, len(rects)):
        area = rects[k][2] * rects[k][3]
        if area > max_area:
            max_area_idx, max_area = k, area
    return rects[max_area_idx]


def predict_location(heatmap):
    """test.py:52-79 on a uint8 (H, W) map -> (x, y, w, h) of the max-area bounding box."""
    if np.amax(heatmap) == 0:
        return 0, 0, 0, 0
    rects = connected_boxes(heatmap)
    if TIE_LAST_WINS:
        rects = rects[::-1]          # cv2 contour order: last discovered first
    max_area_idx = 0
    max_area = rects[0][2] * rects[0][3]
    for i in range(1, len(rects)):
        area = rects[i][2] * rects[i][3]
        if area > max_area:
            max_area_idx = i
            max_area = area
    x, y, w, h, _ = rects[max_area_idx]
    return x, y, w, h


def predict(indices, y_pred=None, c_pred=None, img_scaler=(1, 1)):
    """predict.py:28-69.  indices (N, L, 2) ints; y_pred (N, L, H, W) float; c_pred (N, L, 2)."""
    pred_dict = {"Frame": [], "X": [], "Y": [], "Visibility": []}
    indices = np.asarray(indices)
    batch_size, seq_len = indices.shape[0], indices.shape[1]
    if y_pred is not None:
        y_pred = np.asarray(y_pred) > 0.5
    if c_pred is not None:
        c_pred = np.asarray(c_pred)
    prev_f_i = -1
    for n in range(batch_size):
        for f in range(seq_len):
            f_i = indices[n][f][1]
            if f_i != prev_f_i:
                if c_pred is not None:
                    c_p = c_pred[n][f]
                    # float64 products: under the reference's pinned numpy 1.22.4 an np.float32 scalar times a Python
                    # scalar promotes to float64 (numpy >= 2 would keep float32 and truncate differently)
                    cx_pred = int(float(c_p[0]) * WIDTH * img_scaler[0])
                    cy_pred = int(float(c_p[1]) * HEIGHT * img_scaler[1])
                elif y_pred is not None:
                    bbox = predict_location(to_img(y_pred[n][f]))
                    cx_pred, cy_pred = int(bbox[0] + bbox[2] / 2), int(bbox[1] + bbox[3] / 2)
                    cx_pred, cy_pred = int(cx_pred * img_scaler[0]), int(cy_pred * img_scaler[1])
                else:
                    raise ValueError("Invalid input")
                vis_pred = 0 if cx_pred == 0 and cy_pred == 0 else 1
                pred_dict["Frame"].append(int(f_i))
                pred_dict["X"].append(cx_pred)
                pred_dict["Y"].append(cy_pred)
                pred_dict["Visibility"].append(vis_pred)
                prev_f_i = f_i
            else:
                break
    return pred_dict


PRED_TYPES = ("TP", "TN", "FP1", "FP2", "FN")          # test.py:20-21: Type is the index into this list


def _classify(pred_present, true_present, pred_xy, true_xy, tolerance):
    """The five-way typing of test.py:136-156 / 170-190 given 'is there a ball' on both sides and integer centres."""
    if not pred_present and not true_present:
        return PRED_TYPES.index("TN")
    if pred_present and not true_present:
        return PRED_TYPES.index("FP2")
    if not pred_present and true_present:
        return PRED_TYPES.index("FN")
    dist = math.sqrt((pred_xy[0] - true_xy[0]) ** 2 + (pred_xy[1] - true_xy[1]) ** 2)
    return PRED_TYPES.index("FP1") if dist > tolerance else PRED_TYPES.index("TP")


def evaluate(indices, y_true=None, y_pred=None, c_true=None, c_pred=None, tolerance=4., img_scaler=(1, 1),
             output_bbox=False, output_gt=False):
    """test.py:81-221 restated.  indices (N, L, 2); heat maps (N, L, H, W) or normalised coordinates (N, L, 2).
    Unlike predict(), the repeated-index stop is per sample and compares the whole (rally, frame) pair."""
    out = {k: [] for k in ("Frame", "X", "Y", "Visibility", "Type", "BBox", "Confidence", "X_GT", "Y_GT", "Visibility_GT")}
    indices = np.asarray(indices)
    n_samples, seq_len = indices.shape[0], indices.shape[1]
    heat = y_true is not None and y_pred is not None
    coor = c_true is not None and c_pred is not None
    if heat:
        assert c_true is None and c_pred is None, "Invalid input"
        y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
        h_pred = y_pred > 0.5
    if coor:
        assert y_true is None and y_pred is None, "Invalid input"
        assert not output_bbox, "Coordinate prediction cannot output detection"
        c_true, c_pred = np.array(c_true, copy=True), np.array(c_pred, copy=True)     # test.py:119-122, in the input's dtype
        for c in (c_true, c_pred):
            c[..., 0] = c[..., 0] * WIDTH
            c[..., 1] = c[..., 1] * HEIGHT
    if not heat and not coor:
        raise ValueError("Invalid input")
    for n in range(n_samples):
        prev = (-1, -1)
        for f in range(seq_len):
            d_i = (int(indices[n][f][0]), int(indices[n][f][1]))
            if d_i == prev:
                break
            if coor:
                c_t, c_p = c_true[n][f], c_pred[n][f]
                true_xy = (int(c_t[0]), int(c_t[1]))
                pred_xy = (int(c_p[0]), int(c_p[1]))
                typ = _classify(np.amax(c_p) > 0, np.amax(c_t) > 0, pred_xy, true_xy, tolerance)
            else:
                bt = predict_location(to_img(y_true[n][f]))
                true_xy = (int(bt[0] + bt[2] / 2), int(bt[1] + bt[3] / 2))
                bp = predict_location(to_img(h_pred[n][f]))
                pred_xy = (int(bp[0] + bp[2] / 2), int(bp[1] + bp[3] / 2))
                conf = np.amax(y_pred[n][f][bp[1]:bp[1] + bp[3], bp[0]:bp[0] + bp[2]]) if max(bp) > 0 else 0.
                typ = _classify(bool(np.amax(h_pred[n][f]) > 0), bool(np.amax(y_true[n][f]) > 0), pred_xy, true_xy, tolerance)
            out["Type"].append(typ)
            out["Frame"].append(d_i[1])
            out["X"].append(int(pred_xy[0] * img_scaler[0]))
            out["Y"].append(int(pred_xy[1] * img_scaler[1]))
            out["Visibility"].append(0 if pred_xy == (0, 0) else 1)
            if output_bbox:
                out["BBox"].append([int(bp[0] * img_scaler[0]), int(bp[1] * img_scaler[1]),
                                    int(bp[2] * img_scaler[0]), int(bp[3] * img_scaler[1])])
                out["Confidence"].append(float(conf))
            if output_gt:
                out["X_GT"].append(int(true_xy[0] * img_scaler[0]))
                out["Y_GT"].append(int(true_xy[1] * img_scaler[1]))
                out["Visibility_GT"].append(0 if true_xy == (0, 0) else 1)
            prev = d_i
    if not output_bbox:
        del out["BBox"], out["Confidence"]
    if not output_gt:
        del out["X_GT"], out["Y_GT"], out["Visibility_GT"]
    return out


def get_metric(TP, TN, FP1, FP2, FN):
    """utils/metric.py:22-46: accuracy, precision, recall, F1, miss rate (0 where a denominator is 0)."""
    total = TP + TN + FP1 + FP2 + FN
    accuracy = (TP + TN) / total if total > 0 else 0
    precision = TP / (TP + FP1 + FP2) if (TP + FP1 + FP2) > 0 else 0
    recall = TP / (TP + FN) if (TP + FN) > 0 else 0
    f1 = 2 * precision * recall / (precision + recall) if (precision + recall) > 0 else 0
    miss_rate = FN / (TP + FN) if (TP + FN) > 0 else 0
    return accuracy, precision, recall, f1, miss_rate


def generate_inpaint_mask(pred_dict, th_h=30):
    """test.py:234-258."""
    y = np.array(pred_dict["Y"])
    vis_pred = np.array(pred_dict["Visibility"])
    inpaint_mask = np.zeros_like(y)
    i = 0
    j = 0
    threshold = th_h
    while j < len(vis_pred):
        while i < len(vis_pred) - 1 and vis_pred[i] == 1:
            i += 1
        j = i
        while j < len(vis_pred) - 1 and vis_pred[j] == 0:
            j += 1
        if j == i:
            break
        elif i == 0 and y[j] > threshold:
            inpaint_mask[:j] = 1
        elif (i > 1 and y[i - 1] > threshold) and (j < len(vis_pred) and y[j] > threshold):
            inpaint_mask[i:j] = 1
        else:
            pass
        i = j
    return inpaint_mask.tolist()


def torch_cpu_sum0(rows):
    """`rows.sum(0)` of a contiguous fp32 (L, *tail) array exactly as torch's CPU sum kernel (aten SumKernel.cpp, unchanged
    between the reference's torch 1.10 and 2.10) adds the L rows -- the temporal ensembles of predict.py:183-186 / 268-271:
      * four or more elements per row (the heat maps; strictly: the columns its vectorised outer sum covers): rows are added
        sequentially, r = (((x0 + x1) + x2) + ...);
      * fewer than four (the (L, 2) coordinates: its scalar `row_sum`, ilp_factor 4): four interleaved partial sums
        p_j = x_j + x_{4+j} + ..., the L % 4 leftover rows into p_0, then r = ((p0 + p1) + p2) + p3.
    Identified against the goldens produced by the reference's own loops (tests/golden/ensemble.npz: all 27 cases bit-equal)."""
    rows = np.asarray(rows, dtype=np.float32)
    n = rows.shape[0]
    tail_elems = int(np.prod(rows.shape[1:])) if rows.ndim > 1 else 1
    if tail_elems >= 4:
        acc = np.zeros(rows.shape[1:], np.float32)
        for k in range(n):
            acc = (acc + rows[k]).astype(np.float32)
        return acc
    part = [np.zeros(rows.shape[1:], np.float32) for _ in range(4)]
    q = n // 4
    for k in range(4 * q):
        part[k % 4] = (part[k % 4] + rows[k]).astype(np.float32)
    for k in range(4 * q, n):
        part[0] = (part[0] + rows[k]).astype(np.float32)
    r = part[0]
    for j in (1, 2, 3):
        r = (r + part[j]).astype(np.float32)
    return r


def ensemble_stream(window_batches, seq_len, eval_mode, num_sample):
    """Literal restatement of the buffer loop predict.py:163-209 (heat maps) and
    predict.py:245-301 (coordinates): ``window_batches`` is an iterable of float32 arrays
    (B, L, *tail) -- the per-window network outputs in sliding-step-1 order.  Yields one
    array per batch: the ensembled per-frame predictions (n_frames_in_batch, *tail),
    including the tail flush after the last window.  Bit-identical to the reference's loops
    (products rounded to fp32, then `torch_cpu_sum0`'s order).
    """
    weight = get_ensemble_weight(seq_len, eval_mode)
    buffer_size = seq_len - 1
    batch_i = np.arange(seq_len)
    frame_i = np.arange(seq_len - 1, -1, -1)
    buf = None
    sample_count = 0
    for y_pred in window_batches:
        y_pred = np.asarray(y_pred, dtype=np.float32)
        tail = y_pred.shape[2:]
        if buf is None:
            buf = np.zeros((buffer_size, seq_len) + tail, dtype=np.float32)
        b_size = y_pred.shape[0]
        buf = np.concatenate((buf, y_pred), axis=0)
        out = []
        wb = weight.reshape((seq_len,) + (1,) * len(tail))
        for b in range(b_size):
            if sample_count < buffer_size:
                e = torch_cpu_sum0(buf[batch_i + b, frame_i]) / np.float32(sample_count + 1)
            else:
                e = torch_cpu_sum0((buf[batch_i + b, frame_i] * wb).astype(np.float32))
            out.append(e.astype(np.float32))
            sample_count += 1
            if sample_count == num_sample:
                pad = np.zeros((buffer_size, seq_len) + tail, dtype=np.float32)
                buf = np.concatenate((buf, pad), axis=0)
                for f in range(1, seq_len):
                    e = torch_cpu_sum0(buf[batch_i + b + f, frame_i]) / np.float32(seq_len - f)
                    out.append(e.astype(np.float32))
        yield np.stack(out, axis=0)
        buf = buf[-buffer_size:]


def inpaint_blend_threshold(coor_inpaint, coor_pred, inpaint_mask):
    """predict.py:225-232: out*m + in*(1-m); zero where both x,y < COOR_TH."""
    out = coor_inpaint * inpaint_mask + coor_pred * (1 - inpaint_mask)
    th = (out[:, :, 0] < COOR_TH) & (out[:, :, 1] < COOR_TH)
    out = out.copy()
    out[th] = 0.0
    return out
