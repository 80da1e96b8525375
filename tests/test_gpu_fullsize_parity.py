"""GPU: parity at BASELINE's FULL sizes.

* training step of TrackNet(27, 8) at 288x512 -- loss, heat maps, BN buffers and all 53 gradients against the fp64 oracle
  evaluated on the box's host CPU (the split-K weight gradients at K = N*H*W = 295 k .. 1.47 M pixels and the Winograd
  dgrad / wgrad at production shapes had only been checked at 32x64 / 64x128);
* per-kernel weight / data gradients at 64->64 and 192->64 @ 288x512 against fp64 torch;
* BASELINE configs[4] end to end with the REAL networks: HIP TrackNet(27,8) -> ensemble -> peak-find -> HIP InpaintNet
  against the oracle flow (pipeline_common.check_real_network_pipeline).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from oracle import nets, prng

pytestmark = pytest.mark.gpu


def T(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(shape, seed, lo, hi))


def rel_err(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


def _host_threads():
    """fp64 convolutions on the host: all SMT threads of a 2-socket box are several times SLOWER than a few dozen."""
    return max(1, min(32, (os.cpu_count() or 2) // 2))


def _report(name, obj):
    from conftest import write_report
    write_report(name, obj)


# Regression thresholds of the training-mode heat maps (bar: 1e-4, BASELINE north_star).  Measured on MI355X, rounds 4-5: the default
# F(4x4) training forward 4.5-4.6e-5 at N = 2 and N = 10, the F(2x2) forward 1.6e-5, torch-fp32 itself 1.7e-5 -- a kernel edit that
# drifts past 1.3x of that is noticed here, long before the bar (tests/test_gpu_precision_sweep.py sweeps seeds and weight scales).
HEAT_BOUND = {"f43": 5.5e-5, "f22": 2.5e-5}      # (f43 = the default: since round 6 the first block in F(2x2), the rest F(4x4): 3.0-4.2e-5 measured)


def _fullsize_train_step(gpu_device, n, tag):
    from tracknetv3_amd.model import TrackNet
    from tracknetv3_amd.utils.metric import WBCELoss
    in_dim, out_dim, h, w, seed = 27, 8, 288, 512, 31
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True)
    m = TrackNet(in_dim, out_dim)
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu_device).train()
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    y = nets.disc_heatmaps(n, out_dim, h, w, seed + 2000)
    # the fp64 and the fp32 host oracle side by side in two worker processes (tests/oracle_pool.py) while the GPU runs the step
    import threading
    import oracle_pool
    box = {}
    spec = dict(kind="train", in_dim=in_dim, out_dim=out_dim, seed=seed, gain=None, var_range=None, n=n, h=h, w=w)
    th = threading.Thread(target=lambda: box.update(o=oracle_pool.run([dict(spec, dtype="float64"), dict(spec, dtype="float32")], workers=2)))
    th.start()
    p = m(x.to(gpu_device))
    loss = WBCELoss(p, y.to(gpu_device))
    loss.backward()
    torch.cuda.synchronize(gpu_device)
    th.join()
    (l64, p64, g64, st64), (l32, p32, g32, _) = box["o"]
    e_loss, e_heat = abs(loss.item() - l64.item()), (p.detach().cpu().double() - p64).abs().max().item()
    assert e_loss <= 2e-5, e_loss
    assert e_heat <= HEAT_BOUND["f22" if tag.endswith("f22fwd") else "f43"], e_heat
    # per-channel sums of the heat maps (a bias the max norm would not show)
    ch = (p.detach().cpu().double().sum((0, 2, 3)) - p64.sum((0, 2, 3))).abs() / p64.sum((0, 2, 3)).abs().clamp_min(1e-30)
    assert ch.max().item() <= 1e-4, ch
    after = m.state_dict()
    for k, v in st64.items():
        if "num_batches" in k:
            assert int(after[k]) == 1
        else:
            assert torch.allclose(after[k].cpu().double(), v, rtol=2e-4, atol=2e-6), k
    names = list(g64.keys())
    assert len(names) == 53
    params = dict(m.named_parameters())
    mine = np.array([rel_err(params[k].grad.cpu(), g64[k]) for k in names])
    ref = np.array([rel_err(g32[k], g64[k]) for k in names])
    _report(f"fullsize_train_parity_{tag}.json", {"batch": n, "loss_abs_err": e_loss, "heatmap_max_abs_err": e_heat,
                                                  "heatmap_channel_sum_rel_err": float(ch.max().item()),
                                                  "oracle_fp32_heatmap_err": (p32.double() - p64).abs().max().item(),
                                                  "grad_rel_err": {k: [float(a), float(b)] for k, a, b in zip(names, mine, ref)},
                                                  "worst": [names[int(mine.argmax())], float(mine.max()), float(ref.max())],
                                                  "median": [float(np.median(mine)), float(np.median(ref))]})
    # every parameter: within 2x the torch-fp32 oracle's own deviation from fp64 (taken over all parameters) + 2e-4 (measured: 1.3x at the
    # worst tensor, 1.5-1.6x in the median)
    assert mine.max() <= 2 * ref.max() + 2e-4, (names[int(mine.argmax())], mine.max(), ref.max())
    assert np.median(mine) <= 2 * np.median(ref) + 1e-4, (np.median(mine), np.median(ref))
    for k, a, b in zip(names, mine, ref):              # and no single tensor far outside its own fp32 noise
        assert a <= 4 * b + 5e-4, (k, a, b)


def test_tracknet_train_step_288x512_27to8_vs_fp64_oracle(gpu_device, train_fwd):
    """train.py:92-95 (forward in train mode, WBCELoss, backward) at the production shape, batch 2, with either training forward."""
    _fullsize_train_step(gpu_device, 2, "n2_f43fwd" if train_fwd else "n2_f22fwd")


def test_tracknet_train_step_288x512_batch10_vs_fp64_oracle(gpu_device):
    """The same at BASELINE configs[2]'s per-GPU batch of 10 (default settings): the fp64 and fp32 oracles on the host cores take ~2 minutes."""
    _fullsize_train_step(gpu_device, 10, "n10_default")


@pytest.mark.parametrize("case", [(2, 64, 0, 64, 288, 512), (1, 128, 64, 64, 288, 512)], ids=["64to64", "dual192to64"])
def test_wgrad_dgrad_kernels_288x512_vs_fp64(gpu_device, case):
    """The worst-conditioned reductions of the net: K = N*H*W pixels per filter tap at full resolution, through the kernels
    the training step really uses there (Winograd-form wgrad / dgrad for the plain layer, the low-resolution formulation
    for the decoder entry)."""
    from tracknetv3_amd import ops
    n, c0, c1, cout, h, w = case
    d = gpu_device
    dz = T((n, cout, h, w), 7)
    wt = T((cout, c0 + c1, 3, 3), 8, -0.1, 0.1)
    old = torch.get_num_threads()
    torch.set_num_threads(_host_threads())
    try:
        if c1 == 0:
            x = torch.relu(T((n, c0, h, w), 9))
            xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
            F.conv2d(xd, wd, padding=1).backward(dz.double())
            dw = ops.conv3x3_wgrad_wino(x.to(d), dz.to(d)).cpu()
            dw_direct = ops.conv3x3_wgrad(x.to(d), dz.to(d)).cpu()
            dx = ops.conv3x3_wino(dz.to(d), ops.pack_wino_weights(wt.to(d), transpose_flip=True), c0).cpu()
            assert rel_err(dw, wd.grad) <= 2e-5 and rel_err(dw_direct, wd.grad) <= 2e-5, (rel_err(dw, wd.grad), rel_err(dw_direct, wd.grad))
            assert rel_err(dx, xd.grad) <= 5e-6, rel_err(dx, xd.grad)
        else:
            xl = torch.relu(T((n, c0, h // 2, w // 2), 9))
            sk = torch.relu(T((n, c1, h, w), 10))
            xld, skd, wd = xl.double().requires_grad_(True), sk.double().requires_grad_(True), wt.double().requires_grad_(True)
            F.conv2d(torch.cat([nets.upsample2x_nearest(xld), skd], 1), wd, padding=1).backward(dz.double())
            dw = ops.conv3x3_wgrad_up2x(xl.to(d), sk.to(d), dz.to(d)).cpu()
            d_low = ops.dgrad_up2x(dz.to(d), ops.pack_dgrad_up2x_weights(wt.to(d), c0), c0).cpu()
            d_skip = ops.conv3x3_wino(dz.to(d), ops.pack_wino_weights(wt.to(d), c_from=c0, transpose_flip=True), c1).cpu()
            assert rel_err(dw, wd.grad) <= 2e-5, rel_err(dw, wd.grad)
            assert rel_err(d_low, xld.grad) <= 5e-6 and rel_err(d_skip, skd.grad) <= 5e-6, (rel_err(d_low, xld.grad), rel_err(d_skip, skd.grad))
    finally:
        torch.set_num_threads(old)


def test_predict_video_real_networks_vs_oracle_flow_288x512(gpu_device):
    """BASELINE configs[4] with HIP TrackNet(27,8) + HIP InpaintNet on 32 synthetic 288x512 frames (eval_mode 'weight':
    25 TrackNet windows, 17 InpaintNet windows) against the oracle flow with the oracle networks on the host."""
    from pipeline_common import check_real_network_pipeline
    rep = {}
    old = torch.get_num_threads()
    torch.set_num_threads(_host_threads())
    try:
        check_real_network_pipeline(gpu_device, t=32, batch=5, eval_mode="weight", report=rep)
    finally:
        torch.set_num_threads(old)
    _report("e2e_real_network_report.json", rep)
    fs = rep["final_stage"]
    assert rep["near_threshold_frames"] or (fs["coordinates"] == 64 and fs["integer_mismatches"] == 0)
