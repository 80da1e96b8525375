"""Model factory and constants -- the drop-in for the hot-path part of the reference's ``utils/general.py``
(get_model: utils/general.py:46-80; HEIGHT/WIDTH/SIGMA/DELTA_T/COOR_TH: utils/general.py:15-19;
ResumeArgumentParser: utils/general.py:23-42)."""
import math

from ..model import InpaintNet, TrackNet

HEIGHT = 288
WIDTH = 512
SIGMA = 2.5
DELTA_T = 1 / math.sqrt(HEIGHT ** 2 + WIDTH ** 2)
COOR_TH = DELTA_T * 50
IMG_FORMAT = 'png'


class ResumeArgumentParser():
    """Rebuilds the argument namespace from a checkpoint's ``param_dict`` (same keys as the reference)."""

    _KEYS = ('model_name', 'seq_len', 'epochs', 'batch_size', 'optim', 'learning_rate', 'lr_scheduler', 'bg_mode',
             'alpha', 'frame_alpha', 'mask_ratio', 'tolerance', 'resume_training', 'seed', 'save_dir', 'debug',
             'verbose')

    def __init__(self, param_dict):
        for k in self._KEYS:
            setattr(self, k, param_dict[k])


def get_model(model_name, seq_len=None, bg_mode=None):
    """Create a model by name; same channel plan and error behaviour as the reference.

    'TrackNet': bg_mode 'subtract' -> (L, L); 'subtract_concat' -> (4L, L); 'concat' -> (3(L+1), L);
    anything else -> (3L, L).  'InpaintNet' -> InpaintNet().  Otherwise ValueError('Invalid model name.').
    """
    if model_name == 'TrackNet':
        if bg_mode == 'subtract':
            model = TrackNet(in_dim=seq_len, out_dim=seq_len)
        elif bg_mode == 'subtract_concat':
            model = TrackNet(in_dim=seq_len * 4, out_dim=seq_len)
        elif bg_mode == 'concat':
            model = TrackNet(in_dim=(seq_len + 1) * 3, out_dim=seq_len)
        else:
            model = TrackNet(in_dim=seq_len * 3, out_dim=seq_len)
    elif model_name == 'InpaintNet':
        model = InpaintNet()
    else:
        raise ValueError('Invalid model name.')
    return model


_CSV_COLUMNS = ('Frame', 'Visibility', 'X', 'Y')
_CSV_COLUMNS_INPAINT = ('Frame', 'Visibility_GT', 'X_GT', 'Y_GT', 'Visibility', 'X', 'Y', 'Inpaint_Mask')


def write_pred_csv(pred_dict, save_file, save_inpaint_mask=False):
    """Write a prediction dict as the reference's csv wire format (utils/general.py:319-354): columns
    ``Frame,Visibility,X,Y`` -- or, with ``save_inpaint_mask``, the InpaintNet-training layout
    ``Frame,Visibility_GT,X_GT,Y_GT,Visibility,X,Y,Inpaint_Mask`` -- one row per frame, no index column.  The reference
    goes through ``pandas.DataFrame.to_csv(index=False)``; this writes the same bytes without needing pandas on the box
    (tests/test_boundary.py compares with pandas): integer and bool columns -- what the post-process produces -- as pandas
    prints them; a column holding any float is a float column there, so its integers print as ``1.0``, NaN / None as an
    empty field, and an all-float32 column in float32's shortest form."""
    import math
    cols = _CSV_COLUMNS_INPAINT if save_inpaint_mask else _CSV_COLUMNS
    series = [list(pred_dict[c]) for c in cols]
    n = len(series[0])
    if any(len(s) != n for s in series):
        raise ValueError('All arrays must be of the same length')         # pandas' message for ragged columns

    def column(vals):
        kinds = [getattr(getattr(v, 'dtype', None), 'name', type(v).__name__) for v in vals]
        vals = [v.item() if hasattr(v, 'item') else v for v in vals]
        if all(isinstance(v, bool) for v in vals):
            return ['True' if v else 'False' for v in vals]
        if all(isinstance(v, int) and not isinstance(v, bool) for v in vals):
            return [str(v) for v in vals]
        if all(isinstance(v, int) for v in vals):                          # bools mixed with ints: an object column in pandas, str() per element
            return [str(v) for v in vals]
        if not all(v is None or isinstance(v, (int, float)) for v in vals):
            raise TypeError('write_pred_csv: columns must hold numbers')
        f32 = all(k == 'float32' for k in kinds)
        out = []
        for v in vals:
            if v is None or (isinstance(v, float) and math.isnan(v)):
                out.append('')
            elif f32:
                import numpy as np
                out.append(str(np.float32(v)))
            else:
                out.append(repr(float(v)))
        return out

    text = [column(s) for s in series]
    with open(save_file, 'w', newline='') as f:
        f.write(','.join(cols) + '\n')
        for row in zip(*text):
            f.write(','.join(row) + '\n')
