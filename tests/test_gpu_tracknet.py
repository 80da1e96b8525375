"""GPU (-m gpu): parity of the HIP path (through the C ABI) with the oracle / the reference goldens.
Tolerance: heat-map values within 1e-4 (BASELINE.json north_star); most checks are far tighter."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from test_emu_kernels import UP2X_CASES, UP2X_WINO_CASES, WINO_CASES, _dgrad_up2x_case, _pack_view_case, _up2x_case, _up2x_wino_case, _wino_case
from oracle import nets, prng
from test_emu_kernels import CONV_CASES, T, conv_ref

pytestmark = pytest.mark.gpu
HEATMAP_TOL = 1e-4


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cfg%d_c%d+%d_o%d_%dx%d_up%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[7]))
def test_conv3x3_small_cases(gpu_device, case):
    from tracknetv3_amd import ops
    cfg, n, c0, c1, cout, h, w, up0, relu, affine = case
    seed = 100 + CONV_CASES.index(case) * 10
    wt = T((cout, c0 + c1, 3, 3), seed)
    s0 = T((n, c0, h // 2, w // 2) if up0 else (n, c0, h, w), seed + 1)
    s1 = T((n, c1, h, w), seed + 2) if c1 else None
    sc = T((cout,), seed + 3, 0.5, 1.5) if affine else None
    sh = T((cout,), seed + 4, -0.5, 0.5) if affine else None
    mu = T((cout,), seed + 5, -2.0, 2.0) if (affine and cfg % 2 == 0) else None
    d = gpu_device
    g = lambda t: None if t is None else t.to(d)
    y = ops.conv3x3(g(s0), ops.pack_conv3x3_weights(g(wt)), cout, src1=g(s1), mean=g(mu), scale=g(sc), shift=g(sh), up0=up0,
                    relu=relu, cfg=cfg).cpu()
    ref = conv_ref(s0, s1, wt, sc, sh, up0, relu, mu)
    assert (y.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("cfg", list(range(19)))
def test_conv3x3_every_config_on_network_shapes(gpu_device, cfg):
    """Each compiled tile configuration on mid-sized shapes incl. a two-source decoder-entry layer."""
    from tracknetv3_amd import ops
    info = ops.conv3x3_config_info(cfg)
    cout = 2 * info["m_block"]
    d = gpu_device
    for (c0, c1, h, w, up0) in ((64, 0, 72, 128, False), (128, 64, 36, 64, True), (27, 0, 40, 96, False)):
        wt = T((cout, c0 + c1, 3, 3), 5, -0.1, 0.1)
        s0 = T((2, c0, h // 2, w // 2) if up0 else (2, c0, h, w), 6)
        s1 = T((2, c1, h, w), 7) if c1 else None
        sc, sh, mu = T((cout,), 8, 0.5, 1.5), T((cout,), 9, -0.5, 0.5), T((cout,), 10, -1.0, 1.0)
        y = ops.conv3x3(s0.to(d), ops.pack_conv3x3_weights(wt.to(d)), cout, src1=None if s1 is None else s1.to(d),
                        mean=mu.to(d), scale=sc.to(d), shift=sh.to(d), up0=up0, relu=True, cfg=cfg).cpu()
        ref = conv_ref(s0, s1, wt, sc, sh, up0, True, mu)
        assert (y.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item() + 1e-6, (cfg, c0, c1)


@pytest.mark.parametrize("case", UP2X_CASES + [(2, 512, 256, 36, 64), (1, 128, 64, 144, 256), (2, 256, 128, 9, 40)])
def test_conv_up2x_vs_torch(gpu_device, case):
    n, c0, cout, hl, wl = case
    e_up, e_full = _up2x_case(n, c0, cout, hl, wl, gpu_device, c1=c0 // 2 if c0 >= 64 else 16)
    assert e_up <= 3e-6 and e_full <= 3e-6, (e_up, e_full)


@pytest.mark.parametrize("case", UP2X_WINO_CASES + [(2, 512, 256, 36, 64), (1, 128, 64, 144, 256), (3, 256, 128, 72, 128), (10, 128, 64, 144, 256)])
def test_conv_up2x_wino_vs_torch(gpu_device, case):
    """The upsampled half in Winograd form (9 of the 16 GEMMs) against fp64 torch on the materialised upsampled tensor and against
    the class-filter kernel; the last case is the batch-10 network shape (several tiles per persistent workgroup)."""
    e_ref, e_old = _up2x_wino_case(*case, gpu_device)
    assert e_ref <= 3e-6 and e_old <= 4e-6, (e_ref, e_old)


@pytest.mark.parametrize("case", UP2X_WINO_CASES + [(2, 24, 128, 6, 32), (1, 8, 64, 2, 32), (1, 33, 64, 10, 32), (2, 512, 256, 36, 64), (1, 128, 64, 144, 256),
                                  (3, 256, 128, 72, 128), (10, 128, 64, 144, 256)])
def test_conv_up2x_wino43_vs_torch(gpu_device, case):
    """The upsampled half in F(4x4, 3x3) form (25 of the 36 products, Lavin's points) against fp64 torch and the class-filter kernel: small
    shapes with every border case, the three decoder entries of the network, batch 10."""
    e_ref, e_old = _up2x_wino_case(*case, gpu_device, variant=2)
    assert e_ref <= 3e-5 and e_old <= 3e-5, (e_ref, e_old)


@pytest.mark.parametrize("case", UP2X_CASES + [(2, 512, 256, 36, 64), (1, 128, 64, 144, 256), (2, 256, 128, 9, 40)])
def test_dgrad_up2x_vs_autograd(gpu_device, case):
    assert _dgrad_up2x_case(*case, gpu_device) <= 3e-6


@pytest.mark.parametrize("variant", [5, -1], ids=["persistent", "library_pick"])      # (0 / 2 / 3 / 4 / 7: measurement twins of libtnv3_diag.so since ABI 5 / 6)
@pytest.mark.parametrize("case", WINO_CASES + [(2, 256, 256, 72, 128), (1, 512, 512, 36, 64), (1, 64, 64, 288, 512), (3, 27, 64, 8, 192)])
def test_conv3x3_wino_vs_torch(monkeypatch, gpu_device, case, variant):
    from tracknetv3_amd import ops
    from tracknetv3_amd import tuning
    monkeypatch.setattr(tuning, "WINO_VARIANT", variant)      # per-call kernel variant (the C ABI has no process-wide knob)
    e_plain, e_full = _wino_case(*case, gpu_device)
    assert e_plain <= 4e-6 and e_full <= 8e-6, (e_plain, e_full)


WINO6_GPU_CASES = [(2, 16, 128, 8, 64), (1, 24, 128, 12, 96), (3, 20, 256, 4, 32), (2, 64, 128, 144, 256), (2, 256, 256, 72, 128),
                   (10, 512, 512, 36, 64), (1, 128, 256, 72, 128)]


@pytest.mark.parametrize("case", WINO6_GPU_CASES)
def test_conv3x3_wino_128_channel_kernel_vs_torch_and_streaming_kernel(gpu_device, case):
    """Variant 6 (kernels/conv3x3_wino6_mfma.h): against fp64 torch, bit-identical to variant 5 wherever 5 takes the shape (plain,
    affine + addend + ReLU, and the statistics epilogue), and what `variant` -1 picks for Cout % 128 == 0."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    d = gpu_device
    x, wt = torch.relu(T((n, cin, h, w), 191)).to(d), T((cout, cin, 3, 3), 192, -0.3, 0.3).to(d)
    mean, scale, shift, add = T((cout,), 193).to(d), T((cout,), 194, 0.5, 1.5).to(d), T((cout,), 195).to(d), T((n, cout, h, w), 196).to(d)
    assert ops.wino_variant(-1, cin, cout) == 6
    u6 = ops.pack_wino_weights(wt, variant=6)
    got = ops.conv3x3_wino(x, u6, cout, variant=6)
    got_full = ops.conv3x3_wino(x, u6, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=6)
    if n * cin * cout * h * w <= 2 * 256 * 256 * 72 * 128:
        ref = F.conv2d(x.double().cpu(), wt.double().cpu(), padding=1)
        s = ref.abs().max().item()
        assert (got.cpu().double() - ref).abs().max().item() <= 4e-6 * s
    if w % 64 == 0:
        u5 = ops.pack_wino_weights(wt, variant=5)
        assert torch.equal(got, ops.conv3x3_wino(x, u5, cout, variant=5))
        assert torch.equal(got_full, ops.conv3x3_wino(x, u5, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=5))
        z5, s5 = ops.conv3x3_wino_stats(x, u5, cout, addend=add, variant=5)
        z6, s6 = ops.conv3x3_wino_stats(x, u6, cout, addend=add, variant=6)
        assert torch.equal(z5, z6) and torch.allclose(s5.sum(1), s6.sum(1), rtol=1e-12, atol=1e-9)
        assert torch.equal(got, ops.conv3x3_wino(x, ops.pack_wino_weights(wt), cout))            # -1 picks variant 6 and its layout
    for _ in range(2):                                                                            # run-to-run identical
        assert torch.equal(got_full, ops.conv3x3_wino(x, u6, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=6))


def test_product_library_refuses_the_measurement_twins(gpu_device):
    """the product library dispatches, per kernel family, its default plus one fallback per shape class.  The generations that
    were measured and rejected -- the F(2x2) forward kernels 0 / 2 / 3 / 4 / 7, the 32x32x2 F(4x4) kernel (variant 1) and its panel, the
    Winograd weight gradients 0 / 2 / 3 / 4 / 6 / 7, the LDS-DMA staged direct weight gradient -- are refused with a pointer to libtnv3_diag.so."""
    from tracknetv3_amd import _lib, ops
    d = gpu_device
    x, wt = torch.relu(T((1, 64, 8, 64), 391)).to(d), T((64, 64, 3, 3), 392, -0.3, 0.3).to(d)
    dz = T((1, 64, 8, 64), 393).to(d)
    u5 = ops.pack_wino_weights(wt, variant=5)
    for v in (0, 2, 3, 4, 7):
        with pytest.raises(_lib.Tnv3Error, match="libtnv3_diag"):
            ops.conv3x3_wino(x, u5, 64, variant=v)
    with pytest.raises(_lib.Tnv3Error, match="libtnv3_diag"):
        ops.pack_wino43_weights(wt, variant=1)
    with pytest.raises(_lib.Tnv3Error, match="libtnv3_diag"):
        ops.conv3x3_wino43(x, ops.pack_wino43_weights(wt, variant=0), 64, variant=1)
    for v in (0, 2, 3, 4, 6, 7):
        with pytest.raises(_lib.Tnv3Error, match="libtnv3_diag"):
            ops.conv3x3_wgrad_wino(x, dz, variant=v)
    with pytest.raises(_lib.Tnv3Error, match="libtnv3_diag"):
        ops.conv3x3_wgrad(x, dz, variant=1)
    for v in (5, -1):                                          # what stays: the streaming kernel, the library's pick (5 again at 64 output channels)
        assert torch.equal(ops.conv3x3_wino(x, u5, 64, variant=5), ops.conv3x3_wino(x, ops.pack_wino_weights(wt, variant=v), 64, variant=v))
    # ... and the 128-channel form (6) on a shape it takes: the same sums in the same order as kernel 5
    w128 = (torch.rand(128, 64, 3, 3, device=x.device) - 0.5) * 0.1
    y5 = ops.conv3x3_wino(x, ops.pack_wino_weights(w128, variant=5), 128, variant=5)
    for v in (6, -1):
        assert torch.equal(y5, ops.conv3x3_wino(x, ops.pack_wino_weights(w128, variant=v), 128, variant=v)), v


def test_wino_pack_view(gpu_device):
    _pack_view_case(gpu_device)


def test_wino_default_is_the_library_default(gpu_device):
    from tracknetv3_amd import tuning
    assert tuning.WINO_VARIANT == -1               # -1: the library's default (variant 6 for Cout % 128 == 0 and Cin > 8, else 5)


def test_pool_head_pack(gpu_device):
    from tracknetv3_amd import ops
    d = gpu_device
    x = T((3, 64, 72, 128), 5)
    assert torch.equal(ops.maxpool2x2(x.to(d)).cpu(), F.max_pool2d(x, 2, 2))
    for L in (3, 8):
        w, b = T((L, 64, 1, 1), 8 + L, -0.3, 0.3), T((L,), 9 + L)
        y = ops.head1x1_sigmoid(x.to(d), w.to(d), b.to(d)).cpu()
        ref = torch.sigmoid(F.conv2d(x.double(), w.double(), b.double()))
        assert (y.double() - ref).abs().max() <= 1e-6
    w = T((128, 70, 3, 3), 7)
    p = ops.pack_conv3x3_weights(w.to(d)).cpu()[:-64].reshape(96, 9, 128)
    assert torch.equal(p[:70], w.permute(1, 2, 3, 0).reshape(70, 9, 128)) and p[70:].abs().max() == 0


def _load_model(g, device):
    from tracknetv3_amd.model import TrackNet
    in_dim, out_dim, n, h, w, seed, cal = (int(v) for v in g["meta"])
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=bool(cal))
    m = TrackNet(in_dim, out_dim)
    m.load_state_dict(sd, strict=True)
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)
    return m.to(device).eval(), x, sd


@pytest.mark.parametrize("name", ["tracknet_9_3_32x64.npz", "tracknet_27_8_32x64_cal.npz"])
def test_tracknet_eval_vs_reference_golden_full_tensor(gpu_device, name):
    g = np.load(os.path.join(GOLDEN, name))
    m, x, _ = _load_model(g, gpu_device)
    y = m(x.to(gpu_device)).cpu().numpy()
    err = np.abs(y - g["eval_out"]).max()
    assert err <= 1e-5, err


@pytest.mark.parametrize("seq_len,bg_mode", [(1, ""), (1, "subtract"), (3, "subtract"), (3, "subtract_concat"), (2, "concat"), (8, "")])
def test_every_get_model_channel_plan_runs_and_matches_the_oracle(gpu_device, seq_len, bg_mode):
    """get_model's channel plans (utils/general.py:46-80): in_dim 3 / 1 / 3 / 12 / 9 / 24 -- odd, tiny and non-multiple-of-8
    input widths go through the same kernels (K padded with zero filter rows); eval and train-mode forward vs the oracle."""
    from tracknetv3_amd.utils.general import get_model
    m = get_model("TrackNet", seq_len, bg_mode)
    in_dim, out_dim = nets.tracknet_dims(seq_len, bg_mode)
    assert m.in_dim == in_dim and m.out_dim == out_dim
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), 50 + in_dim, calibrated=True)
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu_device)
    x = nets.synth_input((2, in_dim, 16, 32), 77)
    with torch.no_grad():
        ref_eval = nets.tracknet_forward(sd, x, training=False)
        ref_train = nets.tracknet_forward({k: v.clone() for k, v in sd.items()}, x, training=True)
    assert (m.eval()(x.to(gpu_device)).cpu() - ref_eval).abs().max().item() <= 1e-5
    with torch.no_grad():
        assert (m.train()(x.to(gpu_device)).cpu() - ref_train).abs().max().item() <= 5e-5


def test_tracknet_eval_vs_reference_golden_probes_64x128(gpu_device):
    g = np.load(os.path.join(GOLDEN, "tracknet_9_3_64x128_cal.npz"))
    m, x, _ = _load_model(g, gpu_device)
    y = m(x.to(gpu_device)).cpu()
    assert np.abs(y.numpy().reshape(-1)[g["eval_probe_idx"]] - g["eval_probe_val"]).max() <= 2e-5
    np.testing.assert_allclose(y.double().sum(dim=(2, 3)).numpy(), g["eval_chan_sum64"], rtol=2e-5)


def test_tracknet_full_size_288x512_vs_reference_golden(gpu_device):
    """Config-2 geometry (27->8 channels, 288x512): 8192 reference probes, per-channel sums, range."""
    g = np.load(os.path.join(GOLDEN, "tracknet_27_8_288x512.npz"))
    m, x, sd = _load_model(g, gpu_device)
    y = m(x.to(gpu_device)).cpu()
    err = np.abs(y.numpy().reshape(-1)[g["eval_probe_idx"]] - g["eval_probe_val"]).max()
    assert err <= HEATMAP_TOL / 4, err
    np.testing.assert_allclose(y.double().sum(dim=(2, 3)).numpy(), g["eval_chan_sum"], rtol=2e-5)
    assert abs(y.min().item() - float(g["eval_min"])) <= 1e-5 and abs(y.max().item() - float(g["eval_max"])) <= 1e-5
    # full-tensor comparison with the oracle computed here on the host CPU
    with torch.no_grad():
        ref = nets.tracknet_forward(sd, x, training=False)
    assert (y - ref).abs().max().item() <= HEATMAP_TOL / 4


def test_tracknet_batch10_properties(gpu_device):
    """BASELINE config 2 size (N=10): determinism and sample independence (eval mode has no cross-sample term)."""
    g = np.load(os.path.join(GOLDEN, "tracknet_27_8_288x512.npz"))
    m, x1, _ = _load_model(g, gpu_device)
    x = torch.cat([x1] + [nets.synth_input(tuple(x1.shape), 900 + k) for k in range(9)], 0).to(gpu_device)
    y_a = m(x)
    y_b = m(x)
    assert torch.equal(y_a, y_b)
    y_1 = m(x[:1].contiguous())
    assert (y_a[:1] - y_1).abs().max().item() <= 1e-6
    perm = torch.tensor([3, 1, 4, 0, 9, 2, 6, 5, 8, 7], device=gpu_device)
    assert torch.equal(m(x[perm].contiguous()), y_a[perm])
    err = np.abs(y_a[0].cpu().numpy().reshape(-1)[g["eval_probe_idx"]] - g["eval_probe_val"]).max()
    assert err <= HEATMAP_TOL / 4


def test_tracknet_batch_split_over_two_streams_is_bit_identical(gpu_device):
    """The default eval forward splits a batch 6 : 4 over two HIP streams (tuning.INFER_SPLIT): same bits as the one-stream
    forward, also when the caller runs on a side stream of its own, and switched off inside model.no_infer_split()."""
    from tracknetv3_amd import model as M
    from tracknetv3_amd import tuning
    g = np.load(os.path.join(GOLDEN, "tracknet_27_8_288x512.npz"))
    m, x1, _ = _load_model(g, gpu_device)
    x = torch.cat([x1] + [nets.synth_input(tuple(x1.shape), 700 + k) for k in range(6)], 0).to(gpu_device)     # batch 7 -> 4 + 3
    assert tuning.INFER_SPLIT and x.shape[0] >= tuning.INFER_SPLIT_MIN_BATCH
    calls = []
    orig = m._forward_eval
    m._forward_eval = lambda t, out=None: (calls.append(int(t.shape[0])), orig(t, out=out))[1]      # (the parts write their heads into slices of one output: no concatenation pass)
    y_split = m(x)
    assert calls == [3, 4] or calls == [4, 3], calls              # side-stream half is issued first
    calls.clear()
    with M.no_infer_split():
        y_one = m(x)
    assert calls == [7]
    assert torch.equal(y_split, y_one)
    s = torch.cuda.Stream(gpu_device)
    s.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(s):
        y_side = m(x)
    torch.cuda.current_stream(gpu_device).wait_stream(s)
    assert torch.equal(y_side, y_one)
    calls.clear()
    m(x[:2].contiguous())                                          # below the minimum batch: one stream
    assert calls == [2]


@pytest.mark.parametrize("case", [(1, 128, 16, 2, 32), (2, 512, 256, 36, 64), (1, 128, 64, 144, 256), (3, 256, 128, 72, 128)])
def test_dgrad_up2x_wino_vs_autograd(gpu_device, case):
    """The one-GEMM low-resolution data gradient (K = 9 * Cout) against fp64 autograd and the 4x4 stride-2 kernel."""
    from test_emu_kernels import _dgrad_up2x_wino_case
    e_ref, e_old = _dgrad_up2x_wino_case(*case, gpu_device)
    assert e_ref <= 3e-6 and e_old <= 4e-6, (e_ref, e_old)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 16, 64, 8, 64), (2, 20, 128, 16, 64), (2, 27, 64, 288, 512), (2, 64, 64, 288, 512), (3, 128, 128, 144, 256), (2, 256, 256, 72, 128),
                                  (3, 512, 512, 36, 64), (2, 256, 512, 36, 64)])
def test_conv3x3_wino43_vs_torch(gpu_device, case):
    """Winograd F(4x4, 3x3) forward kernel vs fp64 torch: plain, eval epilogue (affine + addend + ReLU), run-to-run identical, and as
    the data gradient (transposed, flipped panel)."""
    from test_emu_kernels import _wino43_case
    _wino43_case(case, gpu_device)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 20, 128, 16, 64), (2, 64, 64, 64, 128), (2, 128, 128, 32, 64)])
def test_conv3x3_wino43_stats_output_is_repeatable_and_equals_the_plain_kernel(gpu_device, case):
    """The statistics variant of the F(4x4) kernel writes the same z as the plain one, bit for bit, on every run, into NaN-prefilled
    outputs.  (It once lost elements 0, 1 of some float4 stores to a v_add_f64 that overwrote the store's data registers one
    instruction later -- nondeterministically, and only in this instantiation: DESIGN 3.1g.)"""
    import torch
    from tracknetv3_amd import ops
    from test_emu_kernels import T
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 491)).to(gpu_device), T((cout, cin, 3, 3), 492, -0.3, 0.3).to(gpu_device)
    u = ops.pack_wino43_weights(wt)
    y = ops.conv3x3_wino43(x, u, cout)
    real_empty = torch.empty

    def nan_empty(*a, **k):
        t = real_empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    first = None
    for _ in range(12):
        ops.torch.empty = nan_empty
        try:
            z, st = ops.conv3x3_wino43_stats(x, u, cout)
        finally:
            ops.torch.empty = real_empty
        assert torch.equal(z, y)
        if first is None:
            first = st.clone()
            zd = z.double()
            assert torch.allclose(st[:, :, 0].sum(1), zd.sum((0, 2, 3)), rtol=1e-9, atol=1e-9 * float(zd.abs().sum()))
            assert torch.allclose(st[:, :, 1].sum(1), (zd * zd).sum((0, 2, 3)), rtol=1e-9, atol=1e-9 * float((zd * zd).sum()))
        else:
            assert torch.equal(st, first)
