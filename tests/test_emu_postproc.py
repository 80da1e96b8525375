"""CPU: post-process kernels (ensemble, threshold + connected components + largest box) through the emulator,
against the oracle restatement and the reference-derived ensemble goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import postproc as opp
from oracle import prng


def _blob_maps():
    rng = np.random.RandomState(11)
    maps = []
    m = np.zeros((24, 40), np.float32); maps.append(m.copy())                       # empty
    m[3:6, 4:9] = 0.9; maps.append(m.copy())                                         # one blob
    m[10:12, 20:26] = 0.7; m[0, 39] = 0.6; m[23, 0] = 0.6; maps.append(m.copy())     # several, border-touching
    t = np.zeros((24, 40), np.float32); t[1:3, 1:3] = 1; t[6:8, 5:7] = 1; t[6:8, 30:32] = 1; maps.append(t)   # 3-way area tie
    d = np.zeros((24, 40), np.float32)
    for k in range(10): d[5 + k, 5 + k] = 1                                          # diagonal: 8-connectivity
    maps.append(d)
    r = np.zeros((24, 40), np.float32); r[2:12, 2:14] = 1; r[4:10, 4:12] = 0; r[6:8, 7:9] = 1; maps.append(r)  # blob nested in a ring
    maps.append(np.ones((24, 40), np.float32))                                       # everything on
    maps.append(np.full((24, 40), 0.5, np.float32))                                  # exactly 0.5 is NOT foreground
    s = np.zeros((24, 40), np.float32); s[::2, ::2] = 1; maps.append(s)              # isolated pixels (all ties)
    u = np.zeros((24, 40), np.float32); u[2:20, 3] = 1; u[19, 3:30] = 1; u[2:20, 29] = 1; u[2, 10:20] = 1; maps.append(u)  # U + bar
    for dens in (0.05, 0.3, 0.5, 0.62, 0.9):
        maps.append((rng.rand(24, 40) < dens).astype(np.float32))
    sp = np.zeros((24, 40), np.float32)                                              # spiral-ish snake: long union chains
    for y in range(0, 24, 2): sp[y, :] = 1
    for y in range(1, 24, 2): sp[y, 39 if (y // 2) % 2 == 0 else 0] = 1
    maps.append(sp)
    return np.stack(maps)


def _wide_maps():
    """Rows longer than a 64-pixel wave segment and not a multiple of it: runs cut by segment boundaries."""
    rng = np.random.RandomState(12)
    h, w = 14, 150
    maps = [np.ones((h, w), np.float32)]
    for dens in (0.02, 0.2, 0.5, 0.7, 0.95):
        maps.append((rng.rand(h, w) < dens).astype(np.float32))
    st = np.zeros((h, w), np.float32); st[::3, 5:140] = 1; st[1::3, 139] = 1; maps.append(st)     # long runs chained at one end
    cb = np.zeros((h, w), np.float32); cb[::2, ::2] = 1; cb[1::2, 1::2] = 1; maps.append(cb)      # checkerboard: all diagonal links
    vs = np.zeros((h, w), np.float32); vs[:, 63:66] = 1; vs[:, 127:129] = 1; maps.append(vs)      # columns across segment edges
    return np.stack(maps)


@pytest.mark.parametrize("maps_fn", [_blob_maps, _wide_maps])
def test_peakfind_emulated_vs_oracle(emu, maps_fn):
    from tracknetv3_amd import ops
    maps = maps_fn()
    for tie in (True, False):
        got = ops.heatmap_peakfind(torch.from_numpy(maps), 0.5, tie_last_wins=tie).numpy()
        opp.TIE_LAST_WINS = tie
        try:
            want = np.array([opp.predict_location(opp.to_img(m > 0.5)) for m in maps])
        finally:
            opp.TIE_LAST_WINS = True
        assert np.array_equal(got, want), (tie, got.tolist(), want.tolist())


def test_predict_and_predict_location_api(emu):
    from tracknetv3_amd import postprocess as pp
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    idx = g["predict_c_idx"]
    a = pp.predict(torch.from_numpy(idx), c_pred=torch.from_numpy(g["predict_c_in"]), img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_c_out"])
    n_c = 240                                             # fp32(X / 1920), fp32(Y / 1080): float64 products (numpy 1.22.4 promotion)
    idx2 = np.zeros((n_c, 1, 2), dtype=np.int64)
    idx2[:, 0, 1] = np.arange(n_c)
    c2 = np.zeros((n_c, 1, 2), dtype=np.float32)
    c2[:, 0, 0] = (np.arange(n_c) * 8 + 3).astype(np.float64) / 1920
    c2[:, 0, 1] = (np.arange(n_c) * 4 + 1).astype(np.float64) / 1080
    a = pp.predict(torch.from_numpy(idx2), c_pred=torch.from_numpy(c2), img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_c64_out"])
    # heat-map path on small maps (the 288x512 case runs on the GPU); compare with the oracle's predict()
    hm = np.zeros((3, 4, 16, 32), np.float32)
    hm[0, 0, 3:6, 4:9] = 0.9; hm[0, 1, 1:3, 1:3] = 0.7; hm[0, 1, 8:10, 20:22] = 0.8; hm[1, 2, 0:3, 0:2] = 0.51; hm[2, 0, 15, 31] = 1.0
    a = pp.predict(idx, y_pred=torch.from_numpy(hm), img_scaler=(3.75, 2.5))
    b = opp.predict(idx, y_pred=hm, img_scaler=(3.75, 2.5))
    assert a == b
    img = np.zeros((10, 12), np.uint8); img[2:5, 3:8] = 255
    assert pp.predict_location(img) == (3, 2, 5, 3)
    assert pp.predict_location(np.zeros((10, 12), np.uint8)) == (0, 0, 0, 0)
    with pytest.raises(ValueError, match="Invalid input"):
        pp.predict(idx)
    with pytest.raises(ValueError, match="Invalid mode"):
        pp.get_ensemble_weight(8, "nope")
    for L in (1, 3, 8, 16):
        for mode in ("average", "weight"):
            assert np.array_equal(pp.get_ensemble_weight(L, mode).numpy(), g[f"ens_w_{mode}_{L}"])
    # generate_inpaint_mask goldens
    vis, yy, out = g["inpaint_mask_vis"], g["inpaint_mask_y"], g["inpaint_mask_out"]
    r = 0
    for c in range(vis.shape[0]):
        n = int((vis[c] >= 0).sum())
        for th in (30, 14.4):
            assert pp.generate_inpaint_mask({"Y": yy[c][:n].tolist(), "Visibility": vis[c][:n].tolist()}, th_h=th) == out[r][:n].tolist()
            r += 1


def test_ensemble_stream_emulated_vs_reference_goldens(emu):
    from tracknetv3_amd import postprocess as pp
    g = np.load(os.path.join(GOLDEN, "ensemble.npz"))
    k = 0
    while f"heat_{k}_meta" in g:
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"heat_{k}_meta"])
        win = prng.uniform((n_win, L, 4, 8), seed)
        es = pp.EnsembleStream(L, "weight" if wmode else "average", n_win)
        outs = [es.push(torch.from_numpy(win[s:s + batch])) for s in range(0, n_win, batch)]
        mine = torch.cat(outs, 0).numpy()
        want = g[f"heat_{k}_ens"]
        assert mine.shape == want.shape, (k, mine.shape, want.shape)
        assert np.array_equal(mine, want), (k, np.abs(mine - want).max())      # the reference's summation order: bit-equal
        k += 1
    assert k == 24
    j = 0
    while f"coor_{j}_meta" in g:                      # coordinate ensemble: blend + threshold, ensemble, threshold
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"coor_{j}_meta"])
        win = prng.uniform((n_win, L, 2), seed)
        cin = prng.uniform((n_win, L, 2), seed + 100)
        cin[prng.uniform((n_win, L), seed + 150) < 0.2] = 0
        msk = (prng.uniform((n_win, L, 1), seed + 200) < 0.4).astype(np.float32)
        blended = pp.inpaint_blend_threshold(torch.from_numpy(win), torch.from_numpy(cin), torch.from_numpy(msk))
        es = pp.EnsembleStream(L, "weight" if wmode else "average", n_win)
        mine = torch.cat([es.push(blended[s:s + batch]) for s in range(0, n_win, batch)], 0)
        th = (mine[:, 0] < pp.COOR_TH) & (mine[:, 1] < pp.COOR_TH)
        mine[th] = 0
        assert np.array_equal(mine.numpy(), g[f"coor_{j}_ens"]), j
        j += 1
    assert j == 3


def test_predict_video_pipeline_emulated_vs_oracle_flow(emu):
    """predict.py's whole flow (windows -> net -> ensemble -> peak-find -> inpaint mask -> InpaintNet -> ensemble ->
    coordinates) on small maps with a synthetic heat-map stub in place of TrackNet."""
    from pipeline_common import check_pipeline
    check_pipeline(torch.device("cpu"), 24, 40, 31, 5, "weight")


def test_evaluate_emulated_vs_reference_golden(emu):
    """postprocess.evaluate (device peak-finds + box maxima, host typing) reproduces the reference's evaluate() dicts."""
    from pipeline_common import check_evaluate_against_golden
    from tracknetv3_amd import postprocess as pp
    from tracknetv3_amd.utils.metric import get_metric
    g = np.load(os.path.join(GOLDEN, "evaluate.npz"))
    check_evaluate_against_golden(pp.evaluate, g)
    for row in g["get_metric"]:
        assert np.array_equal(np.array(get_metric(*(int(v) for v in row[:5])), dtype=np.float64), row[5:])


def test_box_max_emulated(emu):
    from tracknetv3_amd import ops
    rng = np.random.RandomState(2)
    heat = rng.rand(6, 20, 70).astype(np.float32)
    boxes = np.array([[0, 0, 70, 20], [3, 4, 5, 6], [69, 19, 1, 1], [0, 0, 0, 0], [10, 0, 60, 1], [64, 5, 30, 30]], dtype=np.int32)
    got = ops.heatmap_box_max(torch.from_numpy(heat), torch.from_numpy(boxes)).numpy()
    want = [heat[f, y:y + h, x:x + w].max() if w > 0 and h > 0 else 0.0 for f, (x, y, w, h) in enumerate(boxes)]
    assert np.array_equal(got, np.array(want, dtype=np.float32))
    assert np.array_equal(ops.heatmap_box_max(torch.from_numpy(heat)).numpy(), heat.reshape(6, -1).max(1))
    heat[1, 5, 4] = np.nan
    assert np.isnan(ops.heatmap_box_max(torch.from_numpy(heat), torch.from_numpy(boxes)).numpy()[1])
