"""GPU (-m gpu): post-process (ensemble, peak-find) and InpaintNet through the C ABI vs the oracle / goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import nets, prng
from oracle import postproc as opp
from test_emu_postproc import _blob_maps
from test_emu_kernels import CONV1D_MFMA_CASES, _conv1d_case

pytestmark = pytest.mark.gpu


def test_peakfind_wide_maps_bit_exact(gpu_device):
    from tracknetv3_amd import ops
    from test_emu_postproc import _wide_maps
    maps = _wide_maps()
    got = ops.heatmap_peakfind(torch.from_numpy(maps).to(gpu_device), 0.5).cpu().numpy()
    want = np.array([opp.predict_location(opp.to_img(m > 0.5)) for m in maps])
    assert np.array_equal(got, want)


def test_peakfind_designed_maps_bit_exact(gpu_device):
    from tracknetv3_amd import ops
    maps = _blob_maps()
    for tie in (True, False):
        got = ops.heatmap_peakfind(torch.from_numpy(maps).to(gpu_device), 0.5, tie_last_wins=tie).cpu().numpy()
        opp.TIE_LAST_WINS = tie
        try:
            want = np.array([opp.predict_location(opp.to_img(m > 0.5)) for m in maps])
        finally:
            opp.TIE_LAST_WINS = True
        assert np.array_equal(got, want), (tie, got.tolist(), want.tolist())


def test_peakfind_full_size_batch_bit_exact_and_concurrent_unions(gpu_device):
    """288x512 maps, 80 per call (config-2 batch): sparse discs, dense noise (long union chains across workgroups),
    a full frame, a serpentine -- repeated to catch union-find races."""
    from tracknetv3_amd import ops
    rng = np.random.RandomState(5)
    maps = np.zeros((80, 288, 512), np.float32)
    ii, jj = np.meshgrid(np.arange(288), np.arange(512), indexing="ij")
    for f in range(60):
        for _ in range(f % 4):
            cy, cx, r = rng.randint(0, 288), rng.randint(0, 512), rng.uniform(1.5, 6)
            maps[f][(ii - cy) ** 2 + (jj - cx) ** 2 <= r * r] = 0.9
    for f in range(60, 70):
        maps[f] = (rng.rand(288, 512) < [0.3, 0.45, 0.55, 0.6, 0.7][f % 5]).astype(np.float32)
    maps[70] = 1.0
    maps[71, ::2, :] = 1.0
    maps[71, 1::4, 511] = 1.0
    maps[71, 3::4, 0] = 1.0                       # one serpentine component spanning the whole frame
    maps[72, ::2, ::2] = 1.0                      # 36864 isolated pixels: all ties
    for f in range(73, 80):
        maps[f] = (rng.rand(288, 512) < 0.5).astype(np.float32) * (rng.rand(288, 512) + 0.01)
    hm = torch.from_numpy(maps).to(gpu_device)
    got = ops.heatmap_peakfind(hm, 0.5).cpu().numpy()
    for _ in range(3):
        assert np.array_equal(ops.heatmap_peakfind(hm, 0.5).cpu().numpy(), got)
    from scipy import ndimage
    for f in range(80):
        fg = maps[f] > 0.5
        lab, n = ndimage.label(fg, structure=np.ones((3, 3)))
        if n == 0:
            assert got[f].tolist() == [0, 0, 0, 0]
            continue
        sl = ndimage.find_objects(lab)
        boxes = [(s[1].start, s[0].start, s[1].stop - s[1].start, s[0].stop - s[0].start) for s in sl]
        firsts = ndimage.minimum(np.arange(fg.size).reshape(fg.shape), lab, index=np.arange(1, n + 1))
        best = max(range(n), key=lambda k: (boxes[k][2] * boxes[k][3], firsts[k]))     # tie: last discovered wins
        assert tuple(got[f].tolist()) == boxes[best], f
    sel = [0, 1, 2, 3, 61, 70, 71]                                                      # oracle's own BFS on a subset
    want = np.array([opp.predict_location(opp.to_img(maps[f] > 0.5)) for f in sel])
    assert np.array_equal(got[sel], want)


def test_predict_matches_reference_golden(gpu_device):
    from tracknetv3_amd import postprocess as pp
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    idx = g["predict_c_idx"]
    hm = np.zeros((3, 4, 288, 512), dtype=np.float32)
    hm[0, 0, 100:105, 200:207] = 0.9
    hm[0, 1, 10:12, 10:12] = 0.7
    hm[0, 1, 50:53, 300:303] = 0.8
    hm[1, 2, 0:3, 0:2] = 0.51
    hm[2, 0, 287, 511] = 1.0
    a = pp.predict(torch.from_numpy(idx), y_pred=torch.from_numpy(hm).to(gpu_device), img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_h_out"])
    a = pp.predict(torch.from_numpy(idx), c_pred=torch.from_numpy(g["predict_c_in"]).to(gpu_device), img_scaler=(3.75, 3.75))
    assert np.array_equal(np.array([a["Frame"], a["X"], a["Y"], a["Visibility"]]), g["predict_c_out"])


def test_ensemble_stream_vs_reference_goldens_and_full_size(gpu_device):
    from tracknetv3_amd import postprocess as pp
    g = np.load(os.path.join(GOLDEN, "ensemble.npz"))
    k = 0
    while f"heat_{k}_meta" in g:
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"heat_{k}_meta"])
        win = prng.uniform((n_win, L, 4, 8), seed)
        es = pp.EnsembleStream(L, "weight" if wmode else "average", n_win)
        mine = torch.cat([es.push(torch.from_numpy(win[s:s + batch]).to(gpu_device)) for s in range(0, n_win, batch)], 0)
        assert np.array_equal(mine.cpu().numpy(), g[f"heat_{k}_ens"]), k        # the reference's own loop: bit-equal
        k += 1
    assert k == 24
    j = 0
    while f"coor_{j}_meta" in g:                      # coordinate ensemble (E = 2: torch's four-partial-sum order): bit-equal
        L, wmode, n_win, batch, seed = (int(v) for v in g[f"coor_{j}_meta"])
        win = prng.uniform((n_win, L, 2), seed)
        cin = prng.uniform((n_win, L, 2), seed + 100)
        cin[prng.uniform((n_win, L), seed + 150) < 0.2] = 0
        msk = (prng.uniform((n_win, L, 1), seed + 200) < 0.4).astype(np.float32)
        bl = pp.inpaint_blend_threshold(torch.from_numpy(win).to(gpu_device), torch.from_numpy(cin).to(gpu_device),
                                        torch.from_numpy(msk).to(gpu_device))
        es = pp.EnsembleStream(L, "weight" if wmode else "average", n_win)
        mine = torch.cat([es.push(bl[s:s + batch]) for s in range(0, n_win, batch)], 0)
        mine[(mine[:, 0] < pp.COOR_TH) & (mine[:, 1] < pp.COOR_TH)] = 0
        assert np.array_equal(mine.cpu().numpy(), g[f"coor_{j}_ens"]), j
        j += 1
    assert j == 3
    # full-size heat maps: the literal buffer-loop restatement (oracle) vs the device stream
    L, n_win, batch = 8, 21, 10
    win = prng.uniform((n_win, L, 288, 512), 99)
    want = np.concatenate(list(opp.ensemble_stream([win[s:s + batch] for s in range(0, n_win, batch)], L, "weight", n_win)), 0)
    es = pp.EnsembleStream(L, "weight", n_win)
    mine = torch.cat([es.push(torch.from_numpy(win[s:s + batch]).to(gpu_device)) for s in range(0, n_win, batch)], 0)
    assert mine.shape[0] == n_win + L - 1 and np.array_equal(mine.cpu().numpy(), want)


def test_inpaintnet_forward(gpu_device):
    from tracknetv3_amd.model import InpaintNet
    g = np.load(os.path.join(GOLDEN, "inpaintnet_6x16.npz"))
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net = net.to(gpu_device).eval()
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis)
    out = net((coor * (1 - mask)).to(gpu_device), mask.to(gpu_device)).cpu()
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-6
    for (n2, L2) in ((3, 5), (9, 24), (4099, 16)):
        c2, m2 = nets.synth_input((n2, L2, 2), 9), (nets.synth_input((n2, L2, 1), 10) < 0.5).float()
        with torch.no_grad():
            ref = nets.inpaintnet_forward(sd, c2, m2)
        assert (net(c2.to(gpu_device), m2.to(gpu_device)).cpu() - ref).abs().max().item() <= 2e-6
    # int mask as produced by train.py:153 (`.int()`) is accepted like torch.cat's type promotion
    assert (net(c2.to(gpu_device), m2.int().to(gpu_device)).cpu() - ref).abs().max().item() <= 2e-6


@pytest.mark.parametrize("case", CONV1D_MFMA_CASES + [(4099, 256, 128, 128, 1), (70000, 32, 0, 64, 1)])
def test_conv1d_mfma_vs_torch(gpu_device, case):
    assert _conv1d_case(*case, gpu_device) <= 3e-6


def test_inpaintnet_train_step_vs_reference_golden(gpu_device):
    """forward(train) + masked MSE (train.py:159-161) + backward + clip_grad_norm_ + Adam step (train.py:164-166)."""
    from tracknetv3_amd.model import InpaintNet
    g = np.load(os.path.join(GOLDEN, "inpaintnet_6x16.npz"))
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net = net.to(gpu_device).train()
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis).to(gpu_device)
    gt = nets.synth_input((n, L, 2), 504).to(gpu_device)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.zero_grad()
    out = net((coor.to(gpu_device) * (1 - mask)), mask.int())          # train.py:153 passes an int mask
    loss = torch.nn.MSELoss()(out * mask, gt * mask)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-7
    params = dict(net.named_parameters())
    for k, name in enumerate(g["grad_names"]):
        gr = params[str(name)].grad.double().cpu()
        assert abs(gr.sum().item() - g["grad_sums"][k]) <= 2e-4 * g["grad_abs"][k] + 1e-9, name
        assert abs(gr.abs().sum().item() - g["grad_abs"][k]) <= 2e-4 * g["grad_abs"][k] + 1e-9, name
    d = (params["predictor.weight"].grad.cpu() - torch.from_numpy(g["grad_pred_w"])).abs().max().item()
    assert d <= 2e-5 * np.abs(g["grad_pred_w"]).max()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1)
    opt.step()
    # bigger, ragged batch vs fp64 autograd on the oracle
    n2 = 67
    c2, m2, g2 = nets.synth_input((n2, L, 2), 9), (nets.synth_input((n2, L, 1), 10) < 0.5).float(), nets.synth_input((n2, L, 2), 11)
    net2 = InpaintNet()
    net2.load_state_dict(sd, strict=True)
    net2 = net2.to(gpu_device).train()
    o2 = net2(c2.to(gpu_device), m2.to(gpu_device))
    torch.nn.MSELoss()(o2 * m2.to(gpu_device), (g2 * m2).to(gpu_device)).backward()
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    o64 = nets.inpaintnet_forward(sd64, c2.double(), m2.double())
    (((o64 - g2.double()) * m2.double()) ** 2).mean().backward()
    for name, prm in net2.named_parameters():
        ref = sd64[name].grad
        assert (prm.grad.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-10, name


@pytest.mark.parametrize("n", [32, 300])
def test_inpaintnet_fused_training_kernels_vs_layer_kernels_and_fp64_autograd(gpu_device, monkeypatch, n):
    """The three-launch training step (kernels/inpaint_fused_train.h) at the README batch and at a batch with more sequences than
    workgroup slots x 0.5: equal to the per-layer kernels and to fp64 autograd of the oracle to fp32 summation order; run-to-run identical."""
    from test_emu_training import _inpaint_train_grads
    from tracknetv3_amd import inpaint_ops
    from tracknetv3_amd.model import InpaintNet
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 79)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net = net.to(gpu_device).train()
    L = 16
    coor, gt = nets.synth_input((n, L, 2), 601), nets.synth_input((n, L, 2), 602)
    mask = (nets.synth_input((n, L, 1), 603) < 0.4).float()
    cd, md, gd = coor.to(gpu_device), mask.to(gpu_device), gt.to(gpu_device)
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "1")
    out_f, g_f = _inpaint_train_grads(net, cd, md, gd)
    _, g_f2 = _inpaint_train_grads(net, cd, md, gd)
    assert all(torch.equal(g_f[k], g_f2[k]) for k in g_f)
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "0")
    out_l, g_l = _inpaint_train_grads(net, cd, md, gd)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    o64 = nets.inpaintnet_forward(sd64, (coor * (1 - mask)).double(), mask.double())
    (((o64 - gt.double()) * mask.double()) ** 2).mean().backward()
    assert (out_f - out_l).abs().max().item() <= 2e-6 and (out_f.cpu().double() - o64.detach()).abs().max().item() <= 2e-6
    for name in g_f:
        ref = sd64[name].grad
        s = ref.abs().max().item()
        assert (g_f[name].cpu().double() - ref).abs().max().item() <= 2e-5 * s + 1e-10, name
        assert (g_l[name].cpu().double() - ref).abs().max().item() <= 2e-5 * s + 1e-10, name


@pytest.mark.parametrize("eval_mode", ["weight", "average"])
def test_predict_video_pipeline_vs_oracle_flow(gpu_device, eval_mode):
    from pipeline_common import check_pipeline
    check_pipeline(gpu_device, 288, 512, 45, 10, eval_mode)


def test_preprocessing_1080p_bit_exact_vs_pillow_and_numpy(gpu_device):
    """median background + PIL BICUBIC resize + CHW + /255 on the device, at the real geometry."""
    from oracle import preproc as opre
    from tracknetv3_amd import preprocess as pre
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(3)
    t = 9
    fr = rng.randint(0, 256, (t, 1080, 1920, 3)).astype(np.uint8)
    fr[:, 100:300, 200:800] = 255
    fr[:, 500:700, 1000:1500] = 0
    fr[::2, 900:1000, 100:300] = 17                                   # bimodal over time
    d = torch.from_numpy(fr).to(gpu_device)
    f32, u8 = pre.resize_frames(d[:3], want_f32=True, want_u8=True)
    for k in range(3):
        ref = np.array(Image.fromarray(fr[k]).resize(size=(512, 288)))          # live Pillow on this box
        assert np.array_equal(u8[k].cpu().numpy(), ref)
        assert np.array_equal(f32[k].cpu().numpy(), (np.moveaxis(ref, -1, 0).astype(np.float64) / 255.0).astype(np.float32))
    assert np.array_equal(u8[0].cpu().numpy(), opre.resize_bicubic_u8(fr[0], 512, 288))
    med = pre.median_background(d).cpu().numpy()
    assert np.array_equal(med, np.median(fr, 0).astype("uint8"))
    med8 = pre.median_background(d[:8]).cpu().numpy()                  # even T: mean of the two middle values, truncated
    assert np.array_equal(med8, np.median(fr[:8], 0).astype("uint8"))
    frames, medf = pre.preprocess_video(d, "concat")
    assert frames.shape == (t, 3, 288, 512) and medf.shape == (3, 288, 512)
    want = opre.tracknet_input_from_frames(fr, [0], 8, "concat")
    from tracknetv3_amd.pipeline import _assemble, _windows
    x = _assemble(frames, medf, _windows(t, 8, 1, False)[:1], "concat").cpu().numpy()
    assert np.array_equal(x, want)


def test_difference_frame_modes_1080p_bit_exact(gpu_device):
    from oracle import preproc as opre
    from tracknetv3_amd import preprocess as pre
    from tracknetv3_amd.pipeline import _assemble, _windows
    rng = np.random.RandomState(5)
    fr = rng.randint(0, 256, (8, 1080, 1920, 3)).astype(np.uint8)
    fr[:4, 100:400, 100:900] = 0
    fr[4:, 100:400, 100:900] = 255                                    # channel sums above 255: wrap path
    d = torch.from_numpy(fr).to(gpu_device)
    med64 = np.median(fr, 0)
    m2 = pre.median_background(d, doubled=True)
    assert np.array_equal(m2.cpu().numpy().astype(np.float64) / 2.0, med64)
    diff = pre.difference_frames(d[:2], m2).cpu().numpy()[..., 0]
    assert np.array_equal(diff[0], opre.diff_frame_u8(fr[0], med64)) and np.array_equal(diff[1], opre.diff_frame_u8(fr[1], med64))
    for mode, c in (("subtract", 1), ("subtract_concat", 4)):
        frames, med = pre.preprocess_video(d, mode)
        assert med is None and frames.shape == (8, c, 288, 512)
        x = _assemble(frames, None, _windows(8, 8, 1, False), mode).cpu().numpy()
        assert np.array_equal(x, opre.tracknet_input_from_frames(fr, [0], 8, mode))


def test_evaluate_vs_reference_golden(gpu_device):
    from pipeline_common import check_evaluate_against_golden
    from tracknetv3_amd import postprocess as pp
    g = np.load(os.path.join(GOLDEN, "evaluate.npz"))
    check_evaluate_against_golden(pp.evaluate, g, to_dev=lambda a: a.to(gpu_device))


@pytest.mark.parametrize("n", [1, 32, 257, 1100])
def test_inpaintnet_fused_kernel_vs_layer_kernels_and_oracle(gpu_device, n, monkeypatch):
    """InpaintNet.forward as ONE persistent kernel (activations in LDS, filters streamed into registers) against the nine-launch
    path and the oracle; n = 1100 exceeds two workgroups per CU, i.e. the grid-stride loop runs."""
    from tracknetv3_amd import inpaint_ops
    from tracknetv3_amd.model import InpaintNet
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 78)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net = net.to(gpu_device).eval()
    x, m = nets.synth_input((n, 16, 2), 21), (nets.synth_input((n, 16, 1), 22) < 0.4).float()
    monkeypatch.setattr(inpaint_ops, "FUSED", "1")
    fused = net(x.to(gpu_device), m.to(gpu_device)).cpu()
    monkeypatch.setattr(inpaint_ops, "FUSED", "0")
    layered = net(x.to(gpu_device), m.to(gpu_device)).cpu()
    with torch.no_grad():
        ref = nets.inpaintnet_forward(sd, x, m)
    assert (fused - ref).abs().max().item() <= 2e-6 and (fused - layered).abs().max().item() <= 2e-6
