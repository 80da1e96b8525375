"""CPU: the oracle (oracle/) reproduces every committed golden vector (which were produced by importing the
reference -- tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import nets, postproc, prng


def _g(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.mark.parametrize("name", ["tracknet_9_3_32x64.npz", "tracknet_27_8_32x64_cal.npz"])
def test_tracknet_eval_and_train_forward(name):
    g = _g(name)
    in_dim, out_dim, n, h, w, seed, cal = (int(v) for v in g["meta"])
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=bool(cal))
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    y = nets.disc_heatmaps(n, out_dim, h, w, seed + 2000)
    with torch.no_grad():
        p = nets.tracknet_forward(sd, x, training=False)
    assert np.abs(p.numpy() - g["eval_out"]).max() <= 2e-6
    assert abs(nets.wbce_loss(p, y).item() - float(g["eval_loss"])) <= 1e-6
    stats = {}
    with torch.no_grad():
        pt = nets.tracknet_forward(sd, x, training=True, stats_out=stats)
    assert np.abs(pt.numpy() - g["train_out"]).max() <= 5e-5
    assert abs(nets.wbce_loss(pt, y).item() - float(g["train_loss"])) <= 2e-6
    got = np.concatenate([stats[k].numpy().ravel() for k in g["bn_names"]])
    np.testing.assert_allclose(got, g["bn_after"], rtol=1e-4, atol=1e-6)


def test_tracknet_grads_fp64_oracle_vs_golden():
    g = _g("tracknet_9_3_32x64.npz")
    in_dim, out_dim, n, h, w, seed, cal = (int(v) for v in g["meta"])
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=bool(cal))
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    y = nets.disc_heatmaps(n, out_dim, h, w, seed + 2000)
    loss, _, grads, _ = nets.tracknet_train_step_grads(sd, x, y, torch.float64)
    assert abs(loss.item() - float(g["train_loss64"])) < 1e-10
    for k, name in enumerate(g["grad_names"]):
        gr = grads[str(name)].numpy()
        st = g["grad_stats64"][k]
        np.testing.assert_allclose([gr.sum(), np.abs(gr).sum(), np.abs(gr).max()], st[:3], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(gr.reshape(-1)[g["grad_probe_idx"][k]], st[3:], rtol=1e-7, atol=1e-14)


def test_wbce_edge_cases_and_closed_form_gradient():
    g = _g("wbce_edge.npz")
    p, y = torch.from_numpy(g["p"]).requires_grad_(True), torch.from_numpy(g["y"])
    loss = nets.wbce_loss(p, y)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    np.testing.assert_allclose(p.grad.numpy(), g["grad"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(nets.wbce_grad_closed_form(p.detach(), y).numpy(), g["grad"], rtol=1e-5, atol=1e-9)
    assert nets.wbce_loss(p.detach(), y, reduce=False).shape == (1,)


def test_inpaintnet_forward():
    g = _g("inpaintnet_6x16.npz")
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis)
    with torch.no_grad():
        o = nets.inpaintnet_forward(sd, coor * (1 - mask), mask)
    assert np.abs(o.numpy() - g["out"]).max() <= 1e-6


def test_host_logic_goldens():
    g = _g("host_logic.npz")
    plan, k = g["get_model_plan"], 0
    for bg in ("", "subtract", "subtract_concat", "concat", None, "zzz"):
        for L in (1, 3, 8):
            assert tuple(plan[k][1:]) == nets.tracknet_dims(L, bg)
            k += 1
    for L in (1, 2, 3, 5, 8, 16):
        for mode in ("average", "weight"):
            assert np.array_equal(postproc.get_ensemble_weight(L, mode), g[f"ens_w_{mode}_{L}"])
    vis, yy, out = g["inpaint_mask_vis"], g["inpaint_mask_y"], g["inpaint_mask_out"]
    r = 0
    for c in range(vis.shape[0]):
        n = int((vis[c] >= 0).sum())
        for th in (30, 14.4):
            got = postproc.generate_inpaint_mask({"Y": yy[c][:n].tolist(), "Visibility": vis[c][:n].tolist()}, th_h=th)
            assert got == out[r][:n].tolist()
            r += 1
    xm, ym = nets.mixup_injected(nets.synth_input((4, 3, 8, 16), 11), nets.synth_input((4, 2, 8, 16), 12),
                                 g["mixup_lam"], g["mixup_perm"])
    assert np.array_equal(xm.numpy(), g["mixup_x"]) and np.array_equal(ym.numpy(), g["mixup_y"])
    a = postproc.predict(g["predict_c_idx"], c_pred=g["predict_c_in"], img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_c_out"])
    n_c = 240                                             # fp32(X / 1920), fp32(Y / 1080): float64 products (numpy 1.22.4 promotion)
    idx2 = np.zeros((n_c, 1, 2), dtype=np.int64)
    idx2[:, 0, 1] = np.arange(n_c)
    c2 = np.zeros((n_c, 1, 2), dtype=np.float32)
    c2[:, 0, 0] = (np.arange(n_c) * 8 + 3).astype(np.float64) / 1920
    c2[:, 0, 1] = (np.arange(n_c) * 4 + 1).astype(np.float64) / 1080
    a = postproc.predict(idx2, c_pred=c2, img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_c64_out"])
    hm = np.zeros((3, 4, 288, 512), dtype=np.float32)
    hm[0, 0, 100:105, 200:207] = 0.9
    hm[0, 1, 10:12, 10:12] = 0.7
    hm[0, 1, 50:53, 300:303] = 0.8
    hm[1, 2, 0:3, 0:2] = 0.51
    hm[2, 0, 287, 511] = 1.0
    a = postproc.predict(g["predict_c_idx"], y_pred=hm, img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_h_out"])


def test_ensemble_goldens():
    g = _g("ensemble.npz")
    k = 0
    while f"heat_{k}_meta" in g:
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"heat_{k}_meta"])
        win = prng.uniform((n_win, L, 4, 8), seed)
        mine = np.concatenate(list(postproc.ensemble_stream([win[s:s + batch] for s in range(0, n_win, batch)], L,
                                                            "weight" if wmode else "average", n_win)), 0)
        assert np.array_equal(mine, g[f"heat_{k}_ens"]), k          # bit-equal to the reference's own loop (torch CPU)
        k += 1
    assert k == 24
    j = 0
    while f"coor_{j}_meta" in g:                      # coordinate ensemble: blend + threshold, ensemble, threshold
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"coor_{j}_meta"])
        win = prng.uniform((n_win, L, 2), seed)
        cin = prng.uniform((n_win, L, 2), seed + 100)
        cin[prng.uniform((n_win, L), seed + 150) < 0.2] = 0
        msk = (prng.uniform((n_win, L, 1), seed + 200) < 0.4).astype(np.float32)
        bl = postproc.inpaint_blend_threshold(win, cin, msk)
        mine = np.concatenate(list(postproc.ensemble_stream([bl[s:s + batch] for s in range(0, n_win, batch)], L,
                                                            "weight" if wmode else "average", n_win)), 0)
        mine[(mine[:, 0] < postproc.COOR_TH) & (mine[:, 1] < postproc.COOR_TH)] = 0
        assert np.array_equal(mine, g[f"coor_{j}_ens"]), j
        j += 1
    assert j == 3


def test_predict_location_against_scipy_label():
    """Independent cross-check of the component / bounding-box part (cv2 is not available: parity unpinned)."""
    from scipy import ndimage
    rng = np.random.RandomState(3)
    for trial in range(30):
        dens = [0.02, 0.2, 0.5, 0.8][trial % 4]
        img = (rng.rand(24, 40) < dens).astype(np.uint8) * 255
        lab, n = ndimage.label(img, structure=np.ones((3, 3)))
        boxes = sorted((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start)
                       for sl in ndimage.find_objects(lab))
        mine = sorted(b[:4] for b in postproc.connected_boxes(img))
        assert mine == boxes
        if n:
            x, y, w, h = postproc.predict_location(img)
            assert w * h == max(b[2] * b[3] for b in boxes)
    assert postproc.predict_location(np.zeros((8, 8), np.uint8)) == (0, 0, 0, 0)
    # tie rule: equal areas -> the component discovered LAST in raster order wins (cv2 contour order)
    img = np.zeros((10, 10), np.uint8)
    img[1:3, 1:3] = 255
    img[6:8, 5:7] = 255
    assert postproc.predict_location(img) == (5, 6, 2, 2)


def test_evaluate_and_get_metric_vs_reference_golden():
    """oracle.postproc.evaluate / get_metric against the dicts the reference's own evaluate() (test.py:81-221) and
    get_metric (utils/metric.py:22-46) produced in the golden generator."""
    from pipeline_common import check_evaluate_against_golden
    from oracle import postproc as opp
    g = np.load(os.path.join(GOLDEN, "evaluate.npz"))
    check_evaluate_against_golden(lambda idx, **kw: opp.evaluate(idx.numpy(), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in kw.items()}), g)
    for row in g["get_metric"]:
        got = opp.get_metric(*(int(v) for v in row[:5]))
        assert np.array_equal(np.array(got, dtype=np.float64), row[5:])


def test_predict_location_equals_suzuki_abe_border_following():
    """The published algorithm behind cv2.findContours (Suzuki & Abe 1985, Algorithm 2: outermost borders) restated in
    oracle/postproc.py, against the flood-fill restatement: same selected box on the designed / tie-heavy maps of the OpenCV pin
    test, on random maps of every density, and with blobs nested in holes (which border following does not report at all)."""
    from test_cv2_pin import _maps
    n_tie = 0
    for k, m in enumerate(_maps()):
        img = postproc.to_img(m > 0.5)
        assert tuple(postproc.predict_location(img)) == tuple(postproc.predict_location_suzuki(img)), k
        cont = postproc.suzuki_abe_external(img)
        starts = [y * img.shape[1] + x for y, x, _ in cont]
        assert starts == sorted(starts)                                   # discovery order = raster order of the start pixels
        boxes = postproc.connected_boxes(img)
        assert set(starts) <= {b[4] for b in boxes}                       # every outer border starts at a component's first pixel
        areas = sorted((b[2] * b[3] for b in boxes), reverse=True)
        if len(areas) > 1 and areas[0] == areas[1]:
            n_tie += 1                                                    # a real tie: the order decides, and both derivations agree
            assert tuple(postproc.predict_location_suzuki(img, newest_first=False)) != tuple(postproc.predict_location(img))
    assert n_tie >= 5
    rng = np.random.RandomState(11)
    for trial in range(120):
        img = (rng.rand(24, 40) < [0.02, 0.1, 0.3, 0.5, 0.7, 0.9][trial % 6]).astype(np.uint8) * 255
        assert tuple(postproc.predict_location(img)) == tuple(postproc.predict_location_suzuki(img)), trial
    ring = np.zeros((16, 20), np.uint8)
    ring[2:13, 2:15] = 255
    ring[4:11, 4:13] = 0
    ring[6:9, 6:9] = 255                                                  # a blob inside the ring's hole
    assert len(postproc.connected_boxes(ring)) == 2 and len(postproc.suzuki_abe_external(ring)) == 1
    assert tuple(postproc.predict_location(ring)) == tuple(postproc.predict_location_suzuki(ring)) == (2, 2, 13, 11)
