"""write_pred_csv (utils/general.py:322-354 of the reference: pandas.DataFrame(...).to_csv(index=False)) byte for byte."""
import numpy as np
import pytest

from tracknetv3_amd.utils.general import write_pred_csv

pd = pytest.importorskip("pandas")


def _ref_bytes(pred_dict, path, save_inpaint_mask):
    """The reference's own statements (utils/general.py:339-354)."""
    if save_inpaint_mask:
        df = pd.DataFrame({'Frame': pred_dict['Frame'], 'Visibility_GT': pred_dict['Visibility_GT'], 'X_GT': pred_dict['X_GT'],
                           'Y_GT': pred_dict['Y_GT'], 'Visibility': pred_dict['Visibility'], 'X': pred_dict['X'], 'Y': pred_dict['Y'],
                           'Inpaint_Mask': pred_dict['Inpaint_Mask']})
    else:
        df = pd.DataFrame({'Frame': pred_dict['Frame'], 'Visibility': pred_dict['Visibility'], 'X': pred_dict['X'], 'Y': pred_dict['Y']})
    df.to_csv(path, index=False)
    with open(path, 'rb') as f:
        return f.read()


def _mine(pred_dict, path, save_inpaint_mask):
    write_pred_csv(pred_dict, str(path), save_inpaint_mask=save_inpaint_mask)
    with open(path, 'rb') as f:
        return f.read()


def _dicts():
    rng = np.random.RandomState(5)
    n = 37
    base = {'Frame': list(range(n)), 'Visibility': [int(v) for v in rng.randint(0, 2, n)], 'X': [int(v) for v in rng.randint(0, 1920, n)],
            'Y': [int(v) for v in rng.randint(0, 1080, n)], 'Inpaint_Mask': [int(v) for v in rng.randint(0, 2, n)],
            'Visibility_GT': [int(v) for v in rng.randint(0, 2, n)], 'X_GT': [int(v) for v in rng.randint(0, 1920, n)],
            'Y_GT': [int(v) for v in rng.randint(0, 1080, n)]}
    yield "python ints", base
    yield "numpy int64 scalars", {k: [np.int64(v) for v in vals] for k, vals in base.items()}
    yield "numpy int32 arrays", {k: np.array(vals, dtype=np.int32) for k, vals in base.items()}
    mixed = dict(base)
    mixed['X'] = [float('nan') if i % 5 == 0 else v for i, v in enumerate(base['X'])]          # a NaN makes the column float64
    mixed['Y'] = [np.float32(v) / np.float32(7) for v in base['Y']]                            # an all-float32 column
    mixed['X_GT'] = [v + 0.125 for v in base['X_GT']]
    yield "float / NaN columns", mixed
    yield "empty", {k: [] for k in base}


@pytest.mark.parametrize("save_inpaint_mask", [False, True])
def test_write_pred_csv_equals_pandas(tmp_path, save_inpaint_mask):
    for name, d in _dicts():
        want = _ref_bytes(d, tmp_path / "ref.csv", save_inpaint_mask)
        got = _mine(d, tmp_path / "mine.csv", save_inpaint_mask)
        assert got == want, (name, got[:200], want[:200])


def test_write_pred_csv_ragged_columns_raise_like_pandas(tmp_path):
    d = {'Frame': [0, 1], 'Visibility': [1], 'X': [3, 4], 'Y': [5, 6]}
    with pytest.raises(ValueError, match="same length"):
        write_pred_csv(d, str(tmp_path / "x.csv"))
    with pytest.raises(ValueError, match="same length"):
        pd.DataFrame(d)


def test_a_column_mixing_bools_and_ints_prints_like_pandas_object_column(tmp_path):
    """bool is an int, so [1, True] matched neither the all-bool nor the all-int rule and was printed as floats; pandas keeps
    an object column and prints str() per element."""
    d = {'Frame': [0, 1, 2], 'Visibility': [1, True, 0], 'X': [5, 6, 7], 'Y': [False, 3, True]}
    assert _mine(d, tmp_path / "a.csv", False) == _ref_bytes(d, tmp_path / "b.csv", False)
