"""Heat-map / coordinate post-process with the reference's function names, computed on the device.

Drop-ins for: get_ensemble_weight (test.py:25-50), predict_location (test.py:52-79), generate_inpaint_mask
(test.py:223-258), predict (predict.py:14-69), and the temporal-ensemble buffer loops of predict.py:163-209 /
243-301 (as the ``EnsembleStream`` class).  Thresholding, connected components, bounding boxes, largest-box
selection and the ensemble FMA run in libtnv3_hip.so; only integers (and tiny coordinate arrays) reach the host.
The few scalar conversions that the reference does in Python float64 (centre, image scaling, int() truncation)
are done the same way here, on the (frames x 4) integer result.
"""
import math

import numpy as np
import torch

from . import ops
from .utils.general import HEIGHT, WIDTH, COOR_TH  # noqa: F401

TIE_LAST_WINS = True   # OpenCV contour order x strict '>' at test.py:74 (see oracle/postproc.py)


def get_ensemble_weight(seq_len, eval_mode):
    """Uniform ('average') or triangular ('weight') temporal-ensemble weights; ValueError('Invalid mode') otherwise."""
    if eval_mode == 'average':
        weight = torch.ones(seq_len) / seq_len
    elif eval_mode == 'weight':
        weight = torch.ones(seq_len)
        for i in range(math.ceil(seq_len / 2)):
            weight[i] = (i + 1)
            weight[seq_len - i - 1] = (i + 1)
        weight = weight / weight.sum()
    else:
        raise ValueError('Invalid mode')
    return weight


def _device_of(t):
    if torch.is_tensor(t) and t.is_cuda:
        return t.device
    from . import _lib
    if _lib.is_emulator():
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def predict_location(heatmap):
    """(H, W) uint8 / bool / float map (numpy or tensor; non-zero = foreground) -> (x, y, w, h) of the largest box."""
    hm = torch.as_tensor(np.asarray(heatmap) if not torch.is_tensor(heatmap) else heatmap)
    dev = _device_of(hm)
    hm = (hm != 0).to(device=dev, dtype=torch.float32).unsqueeze(0).contiguous()
    x, y, w, h = ops.heatmap_peakfind(hm, threshold=0.5, tie_last_wins=TIE_LAST_WINS)[0].tolist()
    return x, y, w, h


def predict(indices, y_pred=None, c_pred=None, img_scaler=(1, 1)):
    """Coordinates from heat maps (N, L, H, W) or from inpainted coordinates (N, L, 2); same dict as the reference:
    {'Frame': [], 'X': [], 'Y': [], 'Visibility': []}, stopping each sample at the first repeated frame id."""
    pred_dict = {'Frame': [], 'X': [], 'Y': [], 'Visibility': []}
    batch_size, seq_len = indices.shape[0], indices.shape[1]
    indices = indices.detach().cpu().numpy() if torch.is_tensor(indices) else np.asarray(indices)

    boxes = None
    if c_pred is not None:
        c_pred = c_pred.detach().cpu().numpy() if torch.is_tensor(c_pred) else np.asarray(c_pred)
    elif y_pred is not None:
        y_pred = torch.as_tensor(y_pred)
        dev = _device_of(y_pred)
        hm = y_pred.to(device=dev, dtype=torch.float32).reshape(batch_size * seq_len, y_pred.shape[-2], y_pred.shape[-1])
        boxes = ops.heatmap_peakfind(hm.contiguous(), threshold=0.5, tie_last_wins=TIE_LAST_WINS).cpu().numpy()
        boxes = boxes.reshape(batch_size, seq_len, 4)
    else:
        raise ValueError('Invalid input')

    prev_f_i = -1
    for n in range(batch_size):
        for f in range(seq_len):
            f_i = indices[n][f][1]
            if f_i != prev_f_i:
                if c_pred is not None:
                    c_p = c_pred[n][f]
                    # float64 products, whatever numpy is installed: the reference's pinned numpy 1.22.4 promotes
                    # `np.float32 scalar * Python int` to float64 (numpy >= 2 keeps float32: 45 % of X / 1920 truncate differently)
                    cx_pred, cy_pred = int(float(c_p[0]) * WIDTH * img_scaler[0]), int(float(c_p[1]) * HEIGHT * img_scaler[1])
                else:
                    bx, by, bw, bh = (int(v) for v in boxes[n][f])
                    cx_pred, cy_pred = int(bx + bw / 2), int(by + bh / 2)
                    cx_pred, cy_pred = int(cx_pred * img_scaler[0]), int(cy_pred * img_scaler[1])
                vis_pred = 0 if cx_pred == 0 and cy_pred == 0 else 1
                pred_dict['Frame'].append(int(f_i))
                pred_dict['X'].append(cx_pred)
                pred_dict['Y'].append(cy_pred)
                pred_dict['Visibility'].append(vis_pred)
                prev_f_i = f_i
            else:
                break
    return pred_dict


pred_types = ['TP', 'TN', 'FP1', 'FP2', 'FN']                      # test.py:20-21
pred_types_map = {pred_type: i for i, pred_type in enumerate(pred_types)}


def _pred_type(pred_ball, true_ball, cx_pred, cy_pred, cx_true, cy_true, tolerance):
    if not pred_ball and not true_ball:
        return pred_types_map['TN']
    if pred_ball and not true_ball:
        return pred_types_map['FP2']
    if not pred_ball and true_ball:
        return pred_types_map['FN']
    dist = math.sqrt(pow(cx_pred - cx_true, 2) + pow(cy_pred - cy_true, 2))
    return pred_types_map['FP1'] if dist > tolerance else pred_types_map['TP']


def evaluate(indices, y_true=None, y_pred=None, c_true=None, c_pred=None, tolerance=4., img_scaler=(1, 1),
             output_bbox=False, output_gt=False):
    """Per-frame TP / TN / FP1 / FP2 / FN typing with the reference's signature and result dict (test.py:81-221).

    Heat-map inputs (N, L, H, W), values in [0, 1], stay on the device: both peak-finds (ground truth through
    `to_img`, prediction through `> 0.5`), the "is there a ball" maxima and the detection confidence are computed by
    libtnv3_hip.so; only (N*L) x 4 integers and N*L floats reach the host.  Coordinate inputs (N, L, 2) are typed on the
    host as in the reference.  The inputs are not modified (the reference scales c_true / c_pred in place)."""
    pred_dict = {'Frame': [], 'X': [], 'Y': [], 'Visibility': [], 'Type': [], 'BBox': [], 'Confidence': [],
                 'X_GT': [], 'Y_GT': [], 'Visibility_GT': []}
    batch_size, seq_len = indices.shape[0], indices.shape[1]
    indices = indices.detach().cpu().numpy().tolist() if torch.is_tensor(indices) else np.asarray(indices).tolist()
    heat = y_true is not None and y_pred is not None
    coor = c_true is not None and c_pred is not None
    if heat:
        assert c_true is None and c_pred is None, 'Invalid input'
        y_pred = torch.as_tensor(y_pred)
        dev = _device_of(y_pred)
        h, w = int(y_pred.shape[-2]), int(y_pred.shape[-1])
        yp = y_pred.to(device=dev, dtype=torch.float32).reshape(batch_size * seq_len, h, w).contiguous()
        yt = torch.as_tensor(y_true).to(device=dev, dtype=torch.float32).reshape(batch_size * seq_len, h, w).contiguous()
        # to_img(y_t) = (y_t * 255).astype('uint8') is non-zero iff the fp32 product reaches 1
        box_t = ops.heatmap_peakfind(yt * 255.0, threshold=float(np.nextafter(np.float32(1), np.float32(0))),
                                     tie_last_wins=TIE_LAST_WINS)
        box_p = ops.heatmap_peakfind(yp, threshold=0.5, tie_last_wins=TIE_LAST_WINS)
        true_ball = (ops.heatmap_box_max(yt) > 0).cpu().numpy().reshape(batch_size, seq_len)
        conf = ops.heatmap_box_max(yp, box_p).cpu().numpy().reshape(batch_size, seq_len) if output_bbox else None
        box_t = box_t.cpu().numpy().reshape(batch_size, seq_len, 4)
        box_p = box_p.cpu().numpy().reshape(batch_size, seq_len, 4)
    if coor:
        assert y_true is None and y_pred is None, 'Invalid input'
        assert output_bbox == False, 'Coordinate prediction cannot output detection'  # noqa: E712
        c_true = np.array(c_true.detach().cpu().numpy() if torch.is_tensor(c_true) else c_true, copy=True)
        c_pred = np.array(c_pred.detach().cpu().numpy() if torch.is_tensor(c_pred) else c_pred, copy=True)
        for c in (c_true, c_pred):
            c[..., 0] = c[..., 0] * WIDTH
            c[..., 1] = c[..., 1] * HEIGHT

    for n in range(batch_size):
        prev_d_i = [-1, -1]
        for f in range(seq_len):
            d_i = indices[n][f]
            if d_i == prev_d_i:
                break
            if coor:
                c_t, c_p = c_true[n][f], c_pred[n][f]
                cx_true, cy_true = int(c_t[0]), int(c_t[1])
                cx_pred, cy_pred = int(c_p[0]), int(c_p[1])
                typ = _pred_type(np.amax(c_p) > 0, np.amax(c_t) > 0, cx_pred, cy_pred, cx_true, cy_true, tolerance)
            elif heat:
                bt = [int(v) for v in box_t[n][f]]
                bp = [int(v) for v in box_p[n][f]]
                cx_true, cy_true = int(bt[0] + bt[2] / 2), int(bt[1] + bt[3] / 2)
                cx_pred, cy_pred = int(bp[0] + bp[2] / 2), int(bp[1] + bp[3] / 2)
                typ = _pred_type(bp[2] > 0, bool(true_ball[n][f]), cx_pred, cy_pred, cx_true, cy_true, tolerance)
            else:
                raise ValueError('Invalid input')
            pred_dict['Type'].append(typ)
            pred_dict['Frame'].append(int(d_i[1]))
            pred_dict['X'].append(int(cx_pred * img_scaler[0]))
            pred_dict['Y'].append(int(cy_pred * img_scaler[1]))
            pred_dict['Visibility'].append(0 if cx_pred == 0 and cy_pred == 0 else 1)
            if output_bbox:
                pred_dict['BBox'].append([int(bp[0] * img_scaler[0]), int(bp[1] * img_scaler[1]),
                                          int(bp[2] * img_scaler[0]), int(bp[3] * img_scaler[1])])
                pred_dict['Confidence'].append(float(conf[n][f]))
            if output_gt:
                pred_dict['X_GT'].append(int(cx_true * img_scaler[0]))
                pred_dict['Y_GT'].append(int(cy_true * img_scaler[1]))
                pred_dict['Visibility_GT'].append(0 if cx_true == 0 and cy_true == 0 else 1)
            prev_d_i = d_i
    if not output_bbox:
        del pred_dict['BBox'], pred_dict['Confidence']
    if not output_gt:
        del pred_dict['X_GT'], pred_dict['Y_GT'], pred_dict['Visibility_GT']
    return pred_dict


def generate_inpaint_mask(pred_dict, th_h=30):
    """Mask the invisible runs whose neighbours are both below the height threshold (host integer scan)."""
    y = np.array(pred_dict['Y'])
    vis_pred = np.array(pred_dict['Visibility'])
    inpaint_mask = np.zeros_like(y)
    n = len(vis_pred)
    i = j = 0
    while j < n:
        while i < n - 1 and vis_pred[i] == 1:
            i += 1
        j = i
        while j < n - 1 and vis_pred[j] == 0:
            j += 1
        if j == i:
            break
        elif i == 0 and y[j] > th_h:
            inpaint_mask[:j] = 1
        elif (i > 1 and y[i - 1] > th_h) and (j < n and y[j] > th_h):
            inpaint_mask[i:j] = 1
        i = j
    return inpaint_mask.tolist()


class EnsembleStream:
    """Device-resident replacement of the prediction-buffer loops in predict.py:163-209 / 243-301.

    push(batch) takes the network outputs of the next ``B`` sliding windows (B, L, *tail) and returns the ensembled
    predictions of every frame that just became final -- ``B`` frames, plus the ``L-1`` tail frames once the last
    window (``num_sample``-th) has arrived -- as one device tensor (n_frames, *tail).  Only the last L-1 windows are
    retained between calls.
    """

    def __init__(self, seq_len, eval_mode, num_sample):
        self.seq_len, self.num_sample = int(seq_len), int(num_sample)
        self.weight_host = get_ensemble_weight(seq_len, eval_mode)
        self.weight = None
        self.buf = None          # windows [s_base, s_base + n)
        self.s_base = 0
        self.count = 0           # windows seen so far == next frame to emit

    def push(self, y):
        y = y.detach().to(torch.float32).contiguous()
        if self.weight is None:
            self.weight = self.weight_host.to(y.device)
        b = int(y.shape[0])
        self.buf = y if self.buf is None else torch.cat((self.buf, y), 0)
        t0 = self.count
        self.count += b
        n_frames = b + (self.seq_len - 1 if self.count >= self.num_sample else 0)
        out = ops.ensemble_frames(self.buf, self.s_base, self.weight, t0, n_frames, self.num_sample)
        keep = min(self.seq_len - 1, int(self.buf.shape[0]))
        self.s_base += int(self.buf.shape[0]) - keep
        self.buf = self.buf[int(self.buf.shape[0]) - keep:] if keep else None
        if self.buf is None:
            self.s_base = self.count
        return out


def inpaint_blend_threshold(coor_inpaint, coor_pred, inpaint_mask):
    """predict.py:225-232: out*m + in*(1-m), then zero where both x,y < COOR_TH.  (N, L, 2) tensors, tiny."""
    out = coor_inpaint * inpaint_mask + coor_pred * (1 - inpaint_mask)
    th_mask = ((out[:, :, 0] < COOR_TH) & (out[:, :, 1] < COOR_TH))
    out = out.clone()
    out[th_mask] = 0.
    return out
