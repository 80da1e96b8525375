"""CPU: the UNCHANGED HIP kernel headers + dispatch code, executed by the host SIMT emulator (tests/emu), against
the oracle.  This validates tile / lane / LDS indexing and the MFMA fragment mapping without a GPU; the `-m gpu`
tests repeat the same comparisons on the real library."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import nets, prng


def T(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(shape, seed, lo, hi))


def conv_ref(s0, s1, wt, sc, sh, up0, relu, mu=None):
    x = s0.repeat_interleave(2, 2).repeat_interleave(2, 3) if up0 else s0
    if s1 is not None:
        x = torch.cat([x, s1], 1)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    if sc is not None:
        if mu is not None:
            ref = ref - mu.double()[None, :, None, None]
        ref = ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    return ref.relu() if relu else ref


CONV_CASES = [
    # cfg, n, c0, c1, cout, h, w, up0, relu, affine
    (0, 1, 6, 0, 64, 8, 32, False, False, False),
    (0, 2, 9, 0, 64, 12, 40, False, True, True),      # ragged tile edges, odd Cin, 2 samples
    (7, 1, 32, 16, 64, 8, 16, True, True, True),       # two sources, first one nearest-upsampled
    (4, 1, 8, 0, 128, 6, 32, False, False, True),
    (2, 1, 5, 0, 128, 8, 32, False, True, False),
    (3, 1, 4, 0, 128, 8, 32, False, False, False),
    (5, 1, 8, 0, 64, 4, 64, False, False, False),
    (1, 1, 10, 0, 64, 16, 32, False, False, False),
    (6, 1, 8, 0, 256, 8, 32, False, False, False),     # several channel blocks -> XCD block map
    (-1, 1, 32, 32, 192, 4, 8, True, True, True),      # 3 channel blocks -> fallback block map, auto config
    (8, 1, 7, 0, 64, 12, 40, False, True, True),       # 512-thread workgroups
    (9, 1, 32, 32, 128, 6, 32, True, False, True),
    (10, 1, 6, 0, 64, 8, 32, False, True, True),       # 2-step operand prefetch ring
    (11, 1, 8, 0, 128, 5, 33, False, False, False),
    (12, 2, 3, 0, 64, 4, 32, False, True, True),
    (13, 2, 9, 0, 64, 12, 40, False, True, True),      # LDS-DMA staging: ragged edges, odd Cin (zero-tail padding)
    (14, 1, 32, 16, 64, 8, 16, True, True, True),      # LDS-DMA + two sources + upsample
    (15, 1, 8, 0, 128, 6, 32, False, False, True),
    (16, 2, 9, 0, 64, 12, 40, False, True, True),      # 3-stage LDS-DMA pipeline: 2 chunks (short pipeline)
    (16, 1, 40, 0, 64, 4, 32, False, False, False),    #                           5 chunks
    (17, 1, 32, 16, 128, 8, 16, True, True, True),     #                           12 chunks, two sources
    (18, 1, 8, 0, 64, 8, 32, False, False, True),      #                           1 chunk
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cfg%d_c%d+%d_o%d_%dx%d_up%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[7]))
def test_conv3x3_mfma_emulated(emu, case):
    from tracknetv3_amd import ops
    cfg, n, c0, c1, cout, h, w, up0, relu, affine = case
    seed = 100 + CONV_CASES.index(case) * 10
    wt = T((cout, c0 + c1, 3, 3), seed)
    s0 = T((n, c0, h // 2, w // 2) if up0 else (n, c0, h, w), seed + 1)
    s1 = T((n, c1, h, w), seed + 2) if c1 else None
    sc = T((cout,), seed + 3, 0.5, 1.5) if affine else None
    sh = T((cout,), seed + 4, -0.5, 0.5) if affine else None
    mu = T((cout,), seed + 5, -2.0, 2.0) if (affine and cfg % 2 == 0) else None      # mean is optional
    y = ops.conv3x3(s0, ops.pack_conv3x3_weights(wt), cout, src1=s1, mean=mu, scale=sc, shift=sh, up0=up0, relu=relu, cfg=cfg)
    ref = conv_ref(s0, s1, wt, sc, sh, up0, relu, mu)
    assert (y.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item() + 1e-6


def test_weight_packing_layouts(emu):
    from tracknetv3_amd import ops
    w = T((64, 5, 3, 3), 7)
    p = ops.pack_conv3x3_weights(w)
    assert p[-64:].abs().max() == 0                     # zero tail (padding source of the LDS-DMA loader)
    p = p[:-64].reshape(32, 9, 64)
    assert torch.equal(p[:5], w.permute(1, 2, 3, 0).reshape(5, 9, 64))
    assert p[5:].abs().max() == 0
    pt = ops.pack_conv3x3_weights(w, transpose_flip=True)[:-64].reshape(64, 9, 5)     # dgrad: K = Cout, M = Cin, taps flipped
    assert torch.equal(pt, w.flip(2, 3).permute(0, 2, 3, 1).reshape(64, 9, 5))


def test_bn_scale_pool_head_emulated(emu):
    from tracknetv3_amd import ops
    g, b, rm, rv = T((70,), 1, 0.5, 1.5), T((70,), 2), T((70,), 3), T((70,), 4, 0.5, 2.0)
    sc = ops.bn_eval_scale(g, rv)
    assert torch.allclose(sc, g / torch.sqrt(rv + 1e-5), rtol=1e-6)
    x = T((2, 3, 8, 16), 5)
    assert torch.equal(ops.maxpool2x2(x), F.max_pool2d(x, 2, 2))
    x = T((2, 64, 4, 8), 6)
    for L in (3, 8, 11):
        w, bias = T((L, 64, 1, 1), 8 + L, -0.3, 0.3), T((L,), 9 + L)
        y = ops.head1x1_sigmoid(x, w, bias)
        ref = torch.sigmoid(F.conv2d(x.double(), w.double(), bias.double()))
        assert (y.double() - ref).abs().max() <= 1e-6
        z = ops.head1x1_sigmoid(x, w, bias, apply_sigmoid=False)
        assert (z.double() - F.conv2d(x.double(), w.double(), bias.double())).abs().max() <= 1e-5


UP2X_CASES = [(1, 8, 64, 4, 20), (2, 12, 128, 3, 34), (1, 32, 64, 5, 33), (1, 6, 128, 2, 8)]   # (n, c0, cout, h_low, w_low)


def _up2x_case(n, c0, cout, hl, wl, device, c1=16, cfg=-1):
    """Decoder-entry layer two ways: upsample + concat + 3x3 conv in fp64 torch vs conv_up2x (low-res half) + conv3x3(skip, addend)."""
    from tracknetv3_amd import ops
    xl = T((n, c0, hl, wl), 71)
    skip = T((n, c1, 2 * hl, 2 * wl), 72)
    w = T((cout, c0 + c1, 3, 3), 73, -0.3, 0.3)
    up = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref_up = F.conv2d(up.double(), w[:, :c0].double(), padding=1)
    ref = F.conv2d(torch.cat([up, skip], 1).double(), w.double(), padding=1)
    part = ops.conv_up2x(xl.to(device), ops.pack_up2x_weights(w.to(device), c0), cout, cfg=cfg)
    e_up = ((part.cpu().double() - ref_up).abs().max() / ref_up.abs().max()).item()          # relative to the output scale
    full = ops.conv3x3(skip.to(device), ops.pack_conv3x3_weights(w[:, c0:].contiguous().to(device)), cout, addend=part)
    return e_up, ((full.cpu().double() - ref).abs().max() / ref.abs().max()).item()


def _dgrad_up2x_case(n, c0, cout, hl, wl, device):
    """autograd of conv2d(upsample2x(x_low), W[:, :c0]) w.r.t. x_low in fp64 vs ops.dgrad_up2x."""
    from tracknetv3_amd import ops
    c0 = (c0 + 3) // 4 * 4
    xl = T((n, c0, hl, wl), 81).double().requires_grad_(True)
    w = T((cout, c0 + 8, 3, 3), 82, -0.3, 0.3)
    dz = T((n, cout, 2 * hl, 2 * wl), 83)
    up = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    F.conv2d(up, w[:, :c0].double(), padding=1).backward(dz.double())
    got = ops.dgrad_up2x(dz.to(device), ops.pack_dgrad_up2x_weights(w.to(device), c0), c0)
    return ((got.cpu().double() - xl.grad).abs().max() / xl.grad.abs().max()).item()


@pytest.mark.parametrize("case", UP2X_CASES)
def test_dgrad_up2x_emulated_vs_autograd(emu, case):
    assert _dgrad_up2x_case(*case, "cpu") <= 2e-6


@pytest.mark.parametrize("case", UP2X_CASES)
def test_conv_up2x_emulated_vs_torch(emu, case):
    e_up, e_full = _up2x_case(*case, "cpu")
    assert e_up <= 2e-6 and e_full <= 2e-6, (e_up, e_full)


# (n, c0, cout, h_low, w_low): two / three / nine chunks, one / two channel blocks, several tiles per workgroup, borders on all sides
UP2X_WINO_CASES = [(1, 12, 64, 2, 64), (2, 20, 128, 4, 64), (1, 70, 64, 2, 128), (3, 16, 192, 6, 64)]


def _up2x_wino_case(n, c0, cout, hl, wl, device, variant=0):
    """The upsampled half in Winograd form (variant 0: 9 of the 16 F(2x2) GEMMs; 2: 25 of the 36 F(4x4) products) against fp64 torch on the
    materialised upsampled tensor and against the class-filter kernel (conv_up2x)."""
    from tracknetv3_amd import ops
    assert ops.up2x_wino_supported(c0, cout, hl, wl, variant)
    xl = torch.relu(T((n, c0, hl, wl), 71))
    w = T((cout, c0 + 8, 3, 3), 73, -0.3, 0.3)
    up = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(up.double(), w[:, :c0].double(), padding=1)
    u = ops.pack_up2x_wino_weights(w.to(device), c0, variant=variant)
    got = ops.conv_up2x_wino(xl.to(device), u, cout, variant=variant)
    assert torch.equal(got, ops.conv_up2x_wino(xl.to(device), u, cout, variant=variant))          # deterministic
    old = ops.conv_up2x(xl.to(device), ops.pack_up2x_weights(w.to(device), c0), cout)
    s = ref.abs().max()
    return ((got.cpu().double() - ref).abs().max() / s).item(), ((got - old).abs().max().cpu().double() / s).item()


@pytest.mark.parametrize("cus", [256, 8])
@pytest.mark.parametrize("case", UP2X_WINO_CASES)
def test_conv_up2x_wino_emulated_vs_torch(emu, monkeypatch, case, cus):
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))        # 8 CUs: the persistent workgroups walk several tiles each
    e_ref, e_old = _up2x_wino_case(*case, "cpu")
    assert e_ref <= 3e-6 and e_old <= 4e-6, (e_ref, e_old)


# the F(4x4) form (25 of the 36 products, 16x16x4 kernel): + odd chunk counts of the 16-channel steps (128-channel geometry), 32-wide low-resolution
# rows, a low-resolution height that is not a multiple of 4 (a half-empty last tile row of the 64-channel geometry), c0 <= 8
UP2X_WINO43_CASES = UP2X_WINO_CASES + [(2, 24, 128, 6, 32), (1, 8, 64, 2, 32), (3, 40, 256, 4, 96), (1, 33, 64, 10, 32)]


@pytest.mark.parametrize("cus", [256, 2])
@pytest.mark.parametrize("case", UP2X_WINO43_CASES)
def test_conv_up2x_wino43_emulated_vs_torch(emu, monkeypatch, case, cus):
    """Lavin's interpolation points (the point -1 has to be one of them): 1-5e-6 of the output scale."""
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    e_ref, e_old = _up2x_wino_case(*case, "cpu", variant=2)
    assert e_ref <= 1.5e-5 and e_old <= 1.5e-5, (e_ref, e_old)


def test_conv_up2x_wino_unsupported_shapes_are_refused(emu):
    from tracknetv3_amd import _lib, ops
    assert not ops.up2x_wino_supported(8, 64, 4, 64, 0) and not ops.up2x_wino_supported(16, 64, 3, 64, 0) and not ops.up2x_wino_supported(16, 64, 4, 32, 0)
    assert not ops.up2x_wino_supported(16, 96, 4, 64, 0) and not ops.up2x_wino_supported(16, 96, 4, 64, 2) and not ops.up2x_wino_supported(16, 64, 3, 64, 2)
    assert ops.up2x_wino_supported(8, 64, 4, 32, 2) and not ops.up2x_wino_supported(8, 64, 4, 48, 2)
    w = T((64, 24, 3, 3), 73, -0.3, 0.3)
    with pytest.raises(_lib.Tnv3Error):
        ops.conv_up2x_wino(T((1, 16, 4, 32), 1), ops.pack_up2x_wino_weights(w, 16, variant=0), 64, variant=0)
    with pytest.raises(_lib.Tnv3Error):                   # a panel of the other variant
        ops.conv_up2x_wino(T((1, 16, 4, 64), 1), ops.pack_up2x_wino_weights(w, 16, variant=0), 64, variant=2)


# (n, c0, cout, h_low, w_low): two / three / nine chunks of output channels, one / two ci blocks, borders on all sides
DGRAD_UP2X_WINO_CASES = [(1, 128, 16, 2, 32), (2, 128, 24, 4, 32), (1, 256, 70, 2, 64), (3, 128, 16, 6, 32)]


def _dgrad_up2x_wino_case(n, c0, cout, hl, wl, device, variant=0):
    """The low-resolution data gradient (variant 0: one GEMM with K = 9 * Cout; 2: the 25-of-36 F(4x4) form) against fp64 autograd of
    conv2d(upsample2x(x_low), W[:, :c0]) and against the 4x4 stride-2 kernel (dgrad_up2x)."""
    from tracknetv3_amd import ops
    assert ops.dgrad_up2x_wino_supported(c0, cout, hl, wl, variant)
    xl = T((n, c0, hl, wl), 81).double().requires_grad_(True)
    w = T((cout, c0 + 8, 3, 3), 82, -0.3, 0.3)
    dz = T((n, cout, 2 * hl, 2 * wl), 83)
    up = xl.repeat_interleave(2, 2).repeat_interleave(2, 3)
    F.conv2d(up, w[:, :c0].double(), padding=1).backward(dz.double())
    u = ops.pack_dgrad_up2x_wino_weights(w.to(device), c0, variant=variant)
    got = ops.dgrad_up2x_wino(dz.to(device), u, c0, variant=variant)
    assert torch.equal(got, ops.dgrad_up2x_wino(dz.to(device), u, c0, variant=variant))          # deterministic
    old = ops.dgrad_up2x(dz.to(device), ops.pack_dgrad_up2x_weights(w.to(device), c0), c0)
    s = xl.grad.abs().max()
    return ((got.cpu().double() - xl.grad).abs().max() / s).item(), ((got - old).abs().max().cpu().double() / s).item()


@pytest.mark.parametrize("cus", [256, 8])
@pytest.mark.parametrize("case", DGRAD_UP2X_WINO_CASES)
def test_dgrad_up2x_wino_emulated_vs_autograd(emu, monkeypatch, case, cus):
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))        # 8 CUs: the persistent workgroups walk several tiles each
    e_ref, e_old = _dgrad_up2x_wino_case(*case, "cpu")
    assert e_ref <= 3e-6 and e_old <= 4e-6, (e_ref, e_old)


# the F(4x4) form (25 of the 36 products, MODE 2 of the 16x16x4 kernel): + the 64-channel geometry (c0 % 128 != 0), a low-resolution height that is
# not a multiple of 4 (a half-empty last tile row of that geometry), cout not a multiple of 8 / 16 (partial chunks of the 16-channel steps)
DGRAD_UP2X_WINO43_CASES = DGRAD_UP2X_WINO_CASES + [(2, 64, 24, 6, 32), (1, 192, 9, 2, 64), (1, 64, 33, 10, 32), (2, 256, 40, 4, 96)]


@pytest.mark.parametrize("cus", [256, 2])
@pytest.mark.parametrize("case", DGRAD_UP2X_WINO43_CASES)
def test_dgrad_up2x_wino43_emulated_vs_autograd(emu, monkeypatch, case, cus):
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    e_ref, e_old = _dgrad_up2x_wino_case(*case, "cpu", variant=2)
    assert e_ref <= 1e-5 and e_old <= 1e-5, (e_ref, e_old)


def test_dgrad_up2x_wino_panels_of_the_other_variant_are_refused(emu):
    from tracknetv3_amd import _lib, ops
    w = T((16, 136, 3, 3), 82, -0.3, 0.3)
    dz = T((1, 16, 4, 64), 83)
    assert ops.dgrad_up2x_wino_supported(64, 16, 2, 32, 2) and not ops.dgrad_up2x_wino_supported(64, 16, 2, 32, 0)
    assert not ops.dgrad_up2x_wino_supported(32, 16, 2, 32, 2) and not ops.dgrad_up2x_wino_supported(64, 16, 3, 32, 2)
    with pytest.raises(_lib.Tnv3Error):
        ops.dgrad_up2x_wino(dz, ops.pack_dgrad_up2x_wino_weights(w, 128, variant=0), 128, variant=2)
    with pytest.raises(_lib.Tnv3Error):
        ops.dgrad_up2x_wino(dz, ops.pack_dgrad_up2x_wino_weights(w, 128, variant=2), 128, variant=0)


@pytest.mark.parametrize("cfg", [2, 3])
def test_conv_up2x_big_tile_configs_emulated(emu, cfg):
    e_up, e_full = _up2x_case(1, 8, 128, 5, 36, "cpu", cfg=cfg)       # 8-row tiles: ragged in both directions
    assert e_up <= 2e-6 and e_full <= 2e-6, (e_up, e_full)


WINO_CASES = [(1, 8, 64, 4, 64), (2, 12, 128, 8, 64), (1, 70, 64, 4, 128)]   # (n, cin, cout, h, w)


def _wino_case(n, cin, cout, h, w, device):
    from tracknetv3_amd import ops
    x, wt = torch.relu(T((n, cin, h, w), 91)), T((cout, cin, 3, 3), 92, -0.3, 0.3)
    mean, scale, shift = T((cout,), 93), T((cout,), 94, 0.5, 1.5), T((cout,), 95)
    add = T((n, cout, h, w), 96)
    z = F.conv2d(x.double(), wt.double(), padding=1)
    ref_plain = z
    ref_full = torch.relu((z + add.double() - mean.double().view(1, -1, 1, 1)) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    u = ops.pack_wino_weights(wt.to(device))
    got_plain = ops.conv3x3_wino(x.to(device), u, cout).cpu().double()
    got_full = ops.conv3x3_wino(x.to(device), u, cout, mean=mean.to(device), scale=scale.to(device), shift=shift.to(device), relu=True,
                                addend=add.to(device)).cpu().double()
    s = ref_plain.abs().max()
    return ((got_plain - ref_plain).abs().max() / s).item(), ((got_full - ref_full).abs().max() / s).item()


# emulator-only extras: an interior tile column (no horizontal padding), the stem's 27 channels, and 192 output channels
# (three channel blocks: the generic, non-XCD block map)
WINO_EMU_EXTRA = [(1, 27, 64, 12, 192), (1, 16, 192, 4, 64)]


@pytest.mark.parametrize("variant", [3, 5], ids=["balanced", "persistent"])      # (0 / 2 / 4: measurement twins of libtnv3_diag.so since ABI 5)
@pytest.mark.parametrize("case", WINO_CASES + WINO_EMU_EXTRA)
def test_conv3x3_wino_emulated_vs_torch(monkeypatch, emu, case, variant):
    from tracknetv3_amd import ops
    from tracknetv3_amd import tuning
    monkeypatch.setattr(tuning, "WINO_VARIANT", variant)      # per-call kernel variant (the C ABI has no process-wide knob)
    e_plain, e_full = _wino_case(*case, "cpu")
    assert e_plain <= 3e-6 and e_full <= 6e-6, (e_plain, e_full)


def _pack_view_case(device):
    """tnv3_conv3x3_wino_pack_view (channel slice / data-gradient transpose + flip inside the pack kernel) == packing the
    tensor torch would have materialised; bit-exact."""
    from tracknetv3_amd import ops
    w = T((64, 40, 3, 3), 401, -0.5, 0.5).to(device)
    assert torch.equal(ops.pack_wino_weights(w, c_from=0), ops.pack_wino_weights(w.clone()))
    assert torch.equal(ops.pack_wino_weights(w, c_from=16), ops.pack_wino_weights(w[:, 16:].contiguous()))
    assert torch.equal(ops.pack_wino_weights(w, c_from=8, c_count=24), ops.pack_wino_weights(w[:, 8:32].contiguous()))
    assert torch.equal(ops.pack_wino_weights(w, transpose_flip=True), ops.pack_wino_weights(w.flip(2, 3).transpose(0, 1).contiguous()))
    assert torch.equal(ops.pack_wino_weights(w, c_from=16, transpose_flip=True),
                       ops.pack_wino_weights(w[:, 16:].flip(2, 3).transpose(0, 1).contiguous()))
    from tracknetv3_amd import _lib
    with pytest.raises(_lib.Tnv3Error):
        ops.pack_wino_weights(w, c_from=30, c_count=20)


def test_wino_pack_view_emulated(emu):
    _pack_view_case("cpu")


def _pack_multi_case(device):
    """tnv3_conv3x3_wino_pack_multi: a list of panels -- every layout, slices, data-gradient transposes, more panels than one table
    holds -- in one call == the one-panel calls, bit for bit."""
    from tracknetv3_amd import ops
    ws = [T((64, 27, 3, 3), 411, -0.5, 0.5), T((128, 64, 3, 3), 412, -0.5, 0.5), T((64, 192, 3, 3), 413, -0.5, 0.5), T((256, 128, 3, 3), 414, -0.5, 0.5)]
    ws = [w.to(device) for w in ws]
    specs = [(ws[0], 0, False), (ws[1], 0, False), (ws[1], 0, True), (ws[2], 128, False), (ws[2], 128, True), (ws[3], 0, False), (ws[3], 0, True)]
    specs = specs * 7                                             # 49 panels: two tables
    got = ops.pack_wino_weights_multi(specs)
    assert len(got) == len(specs)
    layouts = set()
    for (w, c_from, flip), u in zip(specs, got):
        want = ops.pack_wino_weights(w, c_from=c_from, transpose_flip=flip)
        assert u.shape == want.shape and torch.equal(u, want), (tuple(w.shape), c_from, flip)
        cin, cout = (int(w.shape[0]), int(w.shape[1]) - c_from) if flip else (int(w.shape[1]) - c_from, int(w.shape[0]))
        layouts.add(ops.wino_layout(None, cin, cout))
    assert len(layouts) >= 2                                      # the streaming kernel's and the 128-channel kernel's panel orders
    assert ops.pack_wino_weights_multi([]) == []
    # F(4x4, 3x3) panels ride in the same launch (a fourth element 43 in the spec)
    mixed = [(ws[1], 0, False, 43), (ws[1], 0, True, 43), (ws[2], 128, False, 43), (ws[3], 0, False, 22), (ws[0], 0, False, 43)]
    for (w, c_from, flip, kind), u in zip(mixed, ops.pack_wino_weights_multi(mixed)):
        want = (ops.pack_wino43_weights if kind == 43 else ops.pack_wino_weights)(w, c_from=c_from, transpose_flip=flip)
        assert u.shape == want.shape and torch.equal(u, want), (tuple(w.shape), c_from, flip, kind)


def test_wino_pack_multi_emulated(emu):
    _pack_multi_case("cpu")


def test_argument_errors_are_reported(emu):
    from tracknetv3_amd import ops, _lib
    w = ops.pack_conv3x3_weights(T((64, 4, 3, 3), 1))
    with pytest.raises(_lib.Tnv3Error, match="multiple of 64"):
        ops.conv3x3(T((1, 4, 8, 32), 2), T((32 * 9 * 40 + 64,), 3), 40)
    with pytest.raises(_lib.Tnv3Error, match="unknown config"):
        ops.conv3x3(T((1, 4, 8, 32), 2), w, 64, cfg=99)
    with pytest.raises(_lib.Tnv3Error, match="channel block"):
        ops.conv3x3(T((1, 4, 8, 32), 2), w, 64, cfg=2)          # 128-channel config on Cout = 64
    with pytest.raises(_lib.Tnv3Error):
        ops.maxpool2x2(T((1, 1, 3, 6), 4))


@pytest.mark.slow
def test_tracknet_eval_forward_emulated_vs_golden(emu):
    """Whole TrackNet(9,3).eval() through the product's host code + emulated kernels vs the reference golden."""
    from tracknetv3_amd.model import TrackNet
    g = np.load(os.path.join(GOLDEN, "tracknet_9_3_32x64.npz"))
    in_dim, out_dim, n, h, w, seed, cal = (int(v) for v in g["meta"])
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=bool(cal))
    m = TrackNet(in_dim, out_dim)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = nets.synth_input((n, in_dim, h, w), seed + 1000)[:1, :, :16, :32].contiguous()   # a crop keeps the emulator fast
    with torch.no_grad():
        ref = nets.tracknet_forward(sd, x, training=False)
    y = m(x)
    assert (y - ref).abs().max().item() <= 1e-5


def test_inpaintnet_forward_emulated_vs_golden(emu):
    from tracknetv3_amd.model import InpaintNet
    g = np.load(os.path.join(GOLDEN, "inpaintnet_6x16.npz"))
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net.eval()
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis)
    out = net(coor * (1 - mask), mask)
    assert out.shape == (n, L, 2)
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-6
    # other sequence lengths / ragged batch (positions tiled by 16, sequences by 8)
    for (n2, L2) in ((3, 5), (9, 24)):
        c2, m2 = nets.synth_input((n2, L2, 2), 9), (nets.synth_input((n2, L2, 1), 10) < 0.5).float()
        with torch.no_grad():
            ref = nets.inpaintnet_forward(sd, c2, m2)
        assert (net(c2, m2) - ref).abs().max().item() <= 2e-6


CONV1D_MFMA_CASES = [   # (n, c0, c1, cout, act): every tile configuration of conv1d_mfma.h, ragged batches, two sources
    (19, 32, 0, 64, 1), (5, 64, 0, 128, 1), (9, 128, 0, 256, 0), (3, 256, 128, 128, 1), (17, 64, 32, 32, 2), (1, 8, 8, 32, 1),
    # channel counts that are multiples of 8 but not of 32 take the 8-channel-stage (throughput) configurations at any batch
    (5, 40, 0, 128, 1), (11, 40, 24, 64, 1), (3, 24, 0, 32, 0),
]


def _conv1d_case(n, c0, c1, cout, act, device):
    import torch.nn.functional as F
    from tracknetv3_amd import ops
    x0 = nets.synth_input((n, c0, 16), 40 + n) - 0.5
    x1 = (nets.synth_input((n, c1, 16), 41 + n) - 0.5) if c1 else None
    w = (nets.synth_input((cout, c0 + c1, 3), 42 + cout) - 0.5) * (2.0 / (3 * (c0 + c1)) ** 0.5)
    b = nets.synth_input((cout,), 43) - 0.5
    z = F.conv1d(torch.cat([x0, x1], 1).double() if c1 else x0.double(), w.double(), b.double(), padding=1)
    ref = {0: z, 1: F.leaky_relu(z, 0.01), 2: torch.sigmoid(z)}[act].float()
    got = ops.conv1d_k3(x0.to(device), w.to(device), b.to(device), src1=None if x1 is None else x1.to(device), act=act).cpu()
    return (got - ref).abs().max().item()


@pytest.mark.parametrize("case", CONV1D_MFMA_CASES)
def test_conv1d_mfma_emulated_vs_torch(emu, case):
    assert _conv1d_case(*case, "cpu") <= 3e-6


def test_empty_batches_are_accepted(emu):
    """N = 0 flows through the reference's torch ops; the boundary accepts it too (no launch, empty result)."""
    from tracknetv3_amd import ops
    from tracknetv3_amd.model import InpaintNet, TrackNet
    m = TrackNet(9, 3).eval()
    assert m(torch.zeros((0, 9, 16, 32))).shape == (0, 3, 16, 32)
    net = InpaintNet().eval()
    assert net(torch.zeros((0, 16, 2)), torch.zeros((0, 16, 1))).shape == (0, 16, 2)
    assert ops.heatmap_peakfind(torch.zeros((0, 8, 16))).shape == (0, 4)
    with pytest.raises(ValueError):
        m(torch.zeros((1, 8, 16, 32)))                # wrong channel count
    with pytest.raises(ValueError):
        m(torch.zeros((1, 9, 12, 32)))                # H not divisible by 8
    with pytest.raises(ValueError):
        net(torch.zeros((2, 16, 3)), torch.zeros((2, 16, 1)))


@pytest.mark.parametrize("case", WINO_CASES + WINO_EMU_EXTRA)
def test_conv3x3_wino_persistent_kernel_is_bit_identical_to_the_balanced_kernel(emu, case):
    """Variant 5 (persistent workgroups) performs the same fp32 operations per element in the same order as variant 3: the outputs must
    agree to the last bit.  (Variants 0 / 2 / 4 -- the generations 3 was derived from -- left the product library with ABI 5.)"""
    from tracknetv3_amd import ops, _lib
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 91)), T((cout, cin, 3, 3), 92, -0.3, 0.3)
    u = ops.pack_wino_weights(wt, variant=3)
    want = ops.conv3x3_wino(x, u, cout, variant=3)
    assert ops.wino_layout(3, cin, cout) == 0 and ops.wino_layout(5, cin, cout) == 0
    assert torch.equal(want, ops.conv3x3_wino(x, u, cout, variant=5))                                         # persistent workgroups
    # (the emulator is built with -DTNV3_DIAG: it dispatches the twins, 3 among them; that the PRODUCT build refuses 0 / 2 / 3 / 4 / 7 is
    #  tests/test_kernel_resources.py::test_product_library_carries_no_measurement_twins and tests/test_gpu_tracknet.py's refusal test)


# more tiles than the 8 workgroups the emulated "8-CU device" launches: every persistent workgroup walks 2-4 tiles (XCD-aware
# and generic block maps, entries of the padded grid that hold no tile, a last workgroup with fewer tiles than the others)
WINO_PERSIST_CASES = [(3, 12, 64, 8, 128), (1, 27, 64, 12, 192), (2, 20, 128, 12, 64), (5, 8, 192, 4, 64)]


def test_persistent_tile_walk_matches_the_block_map(emu_lib_path):
    """ConvTileWalk (adds and compares per tile) == conv_block_map (divisions) for every grid / shape combination tried, and
    once a workgroup meets an entry without a tile no later entry of its walk has one (the kernels stop there)."""
    import ctypes
    lib = ctypes.CDLL(emu_lib_path)
    lib.tnv3_emu_tile_walk_check.restype = ctypes.c_long
    assert lib.tnv3_emu_tile_walk_check() > 100000


@pytest.mark.parametrize("case", WINO_PERSIST_CASES)
def test_conv3x3_wino_persistent_kernel_walks_several_tiles_per_workgroup(emu, monkeypatch, case):
    """Variant 5 = variant 3 as persistent workgroups (one per CU, next tile's first DMAs issued before the output transform,
    accumulators started from the MFMA's inline zero): bit-identical to variant 3, with the affine / addend / ReLU epilogue too."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 91)), T((cout, cin, 3, 3), 92, -0.3, 0.3)
    mean, scale, shift, add = T((cout,), 93), T((cout,), 94, 0.5, 1.5), T((cout,), 95), T((n, cout, h, w), 96)
    u = ops.pack_wino_weights(wt, variant=3)
    want = ops.conv3x3_wino(x, u, cout, variant=3)
    want_full = ops.conv3x3_wino(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=3)
    monkeypatch.setenv("TNV3_EMU_CUS", "8")
    assert torch.equal(want, ops.conv3x3_wino(x, u, cout, variant=5))
    assert torch.equal(want_full, ops.conv3x3_wino(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=5))


# variant 6 (kernels/conv3x3_wino6_mfma.h): 128 output channels x (4 x 32 pixels) per workgroup, A operand from the layout-2 panel.
# Cases: several tiles per persistent workgroup (8 emulated CUs), 2 / 3 / many chunks, a ragged last chunk (Cin % 8 != 0), two
# channel blocks, W = 32 (one tile column) and the XCD-aware / generic block maps.
WINO6_CASES = [(2, 16, 128, 8, 64), (1, 24, 128, 12, 96), (3, 20, 256, 4, 32), (1, 64, 128, 8, 128), (2, 12, 384, 8, 32)]


@pytest.mark.parametrize("case", WINO6_CASES)
def test_conv3x3_wino_128_channel_kernel_is_bit_identical_to_the_streaming_kernel(emu, monkeypatch, case):
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 191)), T((cout, cin, 3, 3), 192, -0.3, 0.3)
    mean, scale, shift, add = T((cout,), 193), T((cout,), 194, 0.5, 1.5), T((cout,), 195), T((n, cout, h, w), 196)
    assert ops.wino_variant(-1, cin, cout) == 6 and ops.wino_layout(-1, cin, cout) == 2 and ops.wino_layout(6, cin, cout) == 2
    assert ops.wino_variant(-1, cin, 64) == 5 and ops.wino_variant(-1, 8, cout) == 5 and ops.wino_variant(3, cin, cout) == 3
    u3, u6 = ops.pack_wino_weights(wt, variant=3), ops.pack_wino_weights(wt, variant=6)
    assert u3.numel() == u6.numel() and not torch.equal(u3, u6)
    assert torch.equal(u3.sort().values, u6.sort().values)                      # the same numbers in another order
    want = ops.conv3x3_wino(x, u3, cout, variant=3) if w % 64 == 0 else None
    want_full = ops.conv3x3_wino(x, u3, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=3) if w % 64 == 0 else None
    if want is None:                                                             # W % 64 != 0: variants 2-5 do not take the shape; fp64 torch then
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    for cus in ("8", "256"):
        monkeypatch.setenv("TNV3_EMU_CUS", cus)
        got = ops.conv3x3_wino(x, u6, cout, variant=6)
        got_full = ops.conv3x3_wino(x, u6, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=6)
        if want is not None:
            assert torch.equal(want, got) and torch.equal(want_full, got_full)
            assert torch.equal(got, ops.conv3x3_wino(x, ops.pack_wino_weights(wt), cout))     # and -1 picks it
        else:
            assert (got.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
            full = torch.relu((ref + add.double() - mean.double()[None, :, None, None]) * scale.double()[None, :, None, None]
                              + shift.double()[None, :, None, None])
            assert (got_full.double() - full).abs().max().item() <= 3e-6 * max(full.abs().max().item(), ref.abs().max().item())


# variant 7: the same kernel with a 64-channel x 64-tile workgroup tile (two tile pairs per transform thread, two raw pieces per thread);
# selectable, not picked by -1 (measured 2-5 % slower than variant 5)
WINO7_CASES = [(2, 16, 64, 8, 64), (1, 27, 64, 12, 192), (3, 20, 192, 4, 64), (1, 64, 64, 8, 128)]


@pytest.mark.parametrize("case", WINO7_CASES)
def test_conv3x3_wino_64_channel_form_is_bit_identical_to_the_streaming_kernel(emu, monkeypatch, case):
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 391)), T((cout, cin, 3, 3), 392, -0.3, 0.3)
    mean, scale, shift, add = T((cout,), 393), T((cout,), 394, 0.5, 1.5), T((cout,), 395), T((n, cout, h, w), 396)
    assert ops.wino_variant(-1, cin, cout) == (6 if cout % 128 == 0 and cin > 8 else 5) and ops.wino_layout(7, cin, cout) == 2
    u5, u7 = ops.pack_wino_weights(wt, variant=5), ops.pack_wino_weights(wt, variant=7)
    want = ops.conv3x3_wino(x, u5, cout, variant=5)
    want_full = ops.conv3x3_wino(x, u5, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=5)
    z5, s5 = ops.conv3x3_wino_stats(x, u5, cout, addend=add, variant=5)
    for cus in ("8", "256"):
        monkeypatch.setenv("TNV3_EMU_CUS", cus)
        assert torch.equal(want, ops.conv3x3_wino(x, u7, cout, variant=7))
        assert torch.equal(want_full, ops.conv3x3_wino(x, u7, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, variant=7))
        assert torch.equal(want, ops.conv3x3_wino(x, ops.pack_wino_weights(wt), cout))           # -1 picks it
        z7, s7 = ops.conv3x3_wino_stats(x, u7, cout, addend=add, variant=7)
        assert torch.equal(z5, z7) and torch.equal(s5, s7)                                        # same tiles, same fixed-order folds


def test_conv3x3_wino_128_channel_kernel_statistics_and_data_gradient_pack(emu, monkeypatch):
    """Variant 6's epilogue statistics (tiles of 4 x 32 pixels) sum to the same per-channel totals as variant 5's (4 x 64), and the
    transposed / flipped layout-2 panel (the data gradient's filter) gives the data gradient of the convolution."""
    from tracknetv3_amd import ops
    monkeypatch.setenv("TNV3_EMU_CUS", "8")
    n, cin, cout, h, w = 2, 16, 128, 8, 64
    x, wt = torch.relu(T((n, cin, h, w), 291)), T((cout, cin, 3, 3), 292, -0.3, 0.3)
    add = T((n, cout, h, w), 293)
    z5, s5 = ops.conv3x3_wino_stats(x, ops.pack_wino_weights(wt, variant=5), cout, addend=add, variant=5)
    z6, s6 = ops.conv3x3_wino_stats(x, ops.pack_wino_weights(wt, variant=6), cout, addend=add, variant=6)
    assert torch.equal(z5, z6) and s6.shape[1] == 2 * s5.shape[1]
    assert torch.allclose(s5.sum(1), s6.sum(1), rtol=1e-13, atol=1e-12)
    ref = torch.stack((z6.double().sum((0, 2, 3)), (z6.double() ** 2).sum((0, 2, 3))), 1)
    assert torch.allclose(s6.sum(1), ref, rtol=1e-12, atol=1e-10)
    # data gradient of a 128 -> 16 ... no: of a layer with 128 INPUT channels (dX has 128 channels = the kernel's "Cout")
    cin2, cout2 = 128, 24
    w2, dz = T((cout2, cin2, 3, 3), 294, -0.3, 0.3), T((n, cout2, h, w), 295)
    assert ops.wino_variant(-1, cout2, cin2) == 6
    dx = ops.conv3x3_wino(dz, ops.pack_wino_weights(w2, transpose_flip=True), cin2)
    xd = torch.zeros((n, cin2, h, w), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xd, w2.double(), padding=1).backward(dz.double())
    assert (dx.double() - xd.grad).abs().max().item() <= 3e-6 * xd.grad.abs().max().item()
    assert torch.equal(dx, ops.conv3x3_wino(dz, ops.pack_wino_weights(w2, transpose_flip=True, variant=3), cin2, variant=3))


def test_inpaintnet_fused_kernel_emulated_vs_layer_kernels_and_oracle(emu, monkeypatch):
    """The single persistent kernel (inpaint_fused.h) against the nine-launch path and the oracle: same function, K walked in a
    different order (tap-major), so equal to fp32 rounding."""
    from tracknetv3_amd import inpaint_ops
    from tracknetv3_amd.model import InpaintNet
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 78)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net.eval()
    x, m = nets.synth_input((3, 16, 2), 21), (nets.synth_input((3, 16, 1), 22) < 0.4).float()
    monkeypatch.setattr(inpaint_ops, "FUSED", "1")
    fused = net(x, m)
    monkeypatch.setattr(inpaint_ops, "FUSED", "0")
    layered = net(x, m)
    with torch.no_grad():
        ref = nets.inpaintnet_forward(sd, x, m)
    assert (fused - ref).abs().max().item() <= 2e-6 and (fused - layered).abs().max().item() <= 2e-6
    # the packed parameters follow in-place updates (version counters)
    monkeypatch.setattr(inpaint_ops, "FUSED", "1")
    with torch.no_grad():
        net.up_1.conv.weight.mul_(1.5)
        net.predictor.bias.add_(0.1)
    sd2 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref2 = nets.inpaintnet_forward(sd2, x, m)
    assert (net(x, m) - ref2).abs().max().item() <= 2e-6


# ---- Winograd F(4x4, 3x3) form (kernels/conv3x3_wino43s_mfma.h = variant 0, conv3x3_wino43_mfma.h = variant 1): another factorisation ->
#      compared with fp64 torch, not bit-wise
WINO43_CASES = [(1, 16, 64, 8, 64), (2, 20, 128, 16, 64), (1, 27, 64, 8, 128), (1, 8, 64, 24, 64), (2, 16, 64, 12, 64), (1, 24, 64, 4, 64),
                (1, 64, 64, 8, 64), (1, 64, 128, 12, 64), (3, 9, 64, 16, 128), (3, 17, 128, 12, 128), (2, 8, 256, 8, 64)]      # (n, cin, cout, h, w); H % 8 == 4: half-empty last tile row; Cin % 64 == 0: + the data gradient


def _wino43_case(case, device, tol=2e-5, variant=None):        # F(4x4, 3x3) in fp32: 4e-6 (K = 27 x 9) .. 9e-6 (K = 256 x 9) of the output scale per layer (F(2x2) and direct: 3-5e-7)
    from tracknetv3_amd import ops, tuning
    if variant is not None:
        old = tuning.WINO43_VARIANT
        tuning.WINO43_VARIANT = variant
        try:
            return _wino43_case(case, device, tol)
        finally:
            tuning.WINO43_VARIANT = old
    n, cin, cout, h, w = case
    x, wt = torch.relu(T((n, cin, h, w), 491)).to(device), T((cout, cin, 3, 3), 492, -0.3, 0.3).to(device)
    mean, scale, shift = T((cout,), 493).to(device), T((cout,), 494, 0.5, 1.5).to(device), T((cout,), 495).to(device)
    add = T((n, cout, h, w), 496).to(device)
    assert ops.wino43_supported(cin, cout, h, w)
    u = ops.pack_wino43_weights(wt)
    ref = F.conv2d(x.double().cpu(), wt.double().cpu(), padding=1)
    mag = ref.abs().max().item()
    got = ops.conv3x3_wino43(x, u, cout)
    assert (got.double().cpu() - ref).abs().max().item() <= tol * mag
    full = torch.relu((ref + add.double().cpu() - mean.double().cpu()[None, :, None, None]) * scale.double().cpu()[None, :, None, None]
                      + shift.double().cpu()[None, :, None, None])
    got_full = ops.conv3x3_wino43(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add)
    assert (got_full.double().cpu() - full).abs().max().item() <= tol * max(mag, full.abs().max().item())
    assert torch.equal(got, ops.conv3x3_wino43(x, u, cout))                                   # deterministic
    # the pooled second output (MaxPool2d(2, 2) from the write-out's registers; variant 1: the separate pass): bit-identical to that pass
    y2, pooled = ops.conv3x3_wino43(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True, addend=add, pool=True)
    assert torch.equal(y2, got_full) and torch.equal(pooled, ops.maxpool2x2(got_full))
    # training forward: raw convolution + addend, and BatchNorm's batch statistics from the same launch's epilogue
    z, st = ops.conv3x3_wino43_stats(x, u, cout, addend=add)
    assert (z.double().cpu() - (ref + add.double().cpu())).abs().max().item() <= tol * max(mag, 1.0)
    zd = z.double().cpu()
    assert tuple(st.shape)[0] == cout and st.dtype == torch.float64
    s1, s2 = st.cpu()[:, :, 0].sum(1), st.cpu()[:, :, 1].sum(1)
    assert (s1 - zd.sum((0, 2, 3))).abs().max().item() <= 1e-9 * zd.abs().sum((0, 2, 3)).max().item()
    assert (s2 - (zd * zd).sum((0, 2, 3))).abs().max().item() <= 1e-9 * (zd * zd).sum((0, 2, 3)).max().item()
    z2, st2 = ops.conv3x3_wino43_stats(x, u, cout, addend=add)
    assert torch.equal(z, z2) and torch.equal(st, st2)
    # the data gradient's filter: conv of dZ with the transposed, flipped weight == autograd's dX
    dz = T((n, cout, h, w), 497).to(device)
    xd = x.double().cpu().requires_grad_(True)
    F.conv2d(xd, wt.double().cpu(), padding=1).backward(dz.double().cpu())
    if cin % 64 == 0:                                                  # (the rounding grows with sqrt(K): K = 9 Cout here, and dZ is signed)
        dx = ops.conv3x3_wino43(dz, ops.pack_wino43_weights(wt, transpose_flip=True), cin)
        assert (dx.double().cpu() - xd.grad).abs().max().item() <= tol * max(1.0, (cout / 256.0) ** 0.5) * xd.grad.abs().max().item()


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("case", WINO43_CASES)
def test_conv3x3_wino43_emulated_vs_torch(emu, monkeypatch, case, variant):
    for cus in ("2", "256"):
        monkeypatch.setenv("TNV3_EMU_CUS", cus)
        _wino43_case(case, "cpu", variant=variant)


def test_conv3x3_wino43_skip_half_slice_and_its_data_gradient(emu, monkeypatch):
    """A decoder entry's skip half: the panel of input channels c_from.. (forward) and its transposed, flipped twin (the data gradient
    towards those channels) through the F(4x4) kernel, both variants."""
    from tracknetv3_amd import ops
    monkeypatch.setenv("TNV3_EMU_CUS", "2")
    n, c_up, c_skip, cout, h, w = 1, 8, 64, 64, 8, 64
    wt = T((cout, c_up + c_skip, 3, 3), 511, -0.3, 0.3)
    skip, dz = T((n, c_skip, h, w), 512), T((n, cout, h, w), 513)
    ref = F.conv2d(skip.double(), wt.double()[:, c_up:], padding=1)
    xd = skip.double().requires_grad_(True)
    F.conv2d(xd, wt.double()[:, c_up:], padding=1).backward(dz.double())
    for variant in (0, 1, 2):
        got = ops.conv3x3_wino43(skip, ops.pack_wino43_weights(wt, c_from=c_up, variant=variant), cout, variant=variant)
        assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        dx = ops.conv3x3_wino43(dz, ops.pack_wino43_weights(wt, c_from=c_up, transpose_flip=True, variant=variant), c_skip, variant=variant)
        assert (dx.double() - xd.grad).abs().max().item() <= 2e-5 * xd.grad.abs().max().item()


def _wino43_panel_reference(w, c_from, flip, variant=1):
    """The F(4x4, 3x3) filter panel from its definition (numpy, fp64): U = G g G^T per (output, input) channel with the Toom-Cook G of the
    kernel's interpolation points (0, +-s, +-2s, infinity; s = 3/4) -- row of point p = [1 p p^2] / prod_{q != p} (p - q), times |p| for
    p != 0 (the kernel's A^T has those columns divided by |p|), the row of infinity [0 0 1] -- laid out
    [co / 32][chunk of 8 ci][xg = 3x3 block of the 6x6 xi][x9][lane = (ci % 2) * 32 + co % 32][ci % 8 / 2] (variant 1) or
    [co / 16][chunk][pair = 3 i + j / 2][lane = (ci % 8 / 2) * 16 + co % 16][(ci % 2) * 2 + j % 2] (variant 0), zero for ci >= Cin."""
    sc = 0.75
    pts = [0.0, sc, -sc, 2 * sc, -2 * sc]
    G = np.array([[1.0, p, p * p] for p in pts] + [[0.0, 0.0, 1.0]])
    for k, p in enumerate(pts):
        G[k] *= (abs(p) if p else 1.0) / np.prod([p - q for q in pts if q != p])
    w = w.double().numpy()
    f = w[:, c_from:, ::-1, ::-1].transpose(1, 0, 2, 3) if flip else w[:, c_from:]      # flip: the data gradient's transposed, reversed filter
    cout, cin = f.shape[:2]
    U = np.einsum("ia,ocab,jb->ocij", G, f, G)
    if variant != 1:
        out = np.zeros((cout // 16, (cin + 7) // 8, 18, 64, 4))
        for ci in range(cin):
            g, sx = (ci % 8) // 2, ci % 2
            for i in range(6):
                for j in range(6):
                    out[:, ci // 8, 3 * i + j // 2, g * 16:g * 16 + 16, sx * 2 + j % 2] = U[:, ci, i, j].reshape(cout // 16, 16)
        return out.ravel()
    out = np.zeros((cout // 32, (cin + 7) // 8, 4, 9, 64, 4))
    for ci in range(cin):
        for xg in range(4):
            for x9 in range(9):
                i, j = 3 * (xg >> 1) + x9 // 3, 3 * (xg & 1) + x9 % 3
                out[:, ci // 8, xg, x9, (ci % 2) * 32:(ci % 2) * 32 + 32, (ci % 8) // 2] = U[:, ci, i, j].reshape(cout // 32, 32)
    return out.ravel()


@pytest.mark.parametrize("shape,c_from,flip", [((64, 27, 3, 3), 0, False), ((64, 64, 3, 3), 0, True), ((128, 64, 3, 3), 0, False), ((64, 192, 3, 3), 128, False),
                                               ((64, 192, 3, 3), 128, True), ((64, 96, 3, 3), 37, False), ((96, 64, 3, 3), 0, True), ((64, 20, 3, 3), 3, False)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_wino43_pack_equals_its_definition(emu, shape, c_from, flip, variant):
    """tnv3_conv3x3_wino43_pack (one work item per lane quad: 36 float4 stores) against G g G^T in numpy: input-channel slices, partial last
    chunks, the data gradient's transpose + flip, and the zero tail."""
    from tracknetv3_amd import ops
    w = T(shape, 411, -0.5, 0.5)
    u = ops.pack_wino43_weights(w, c_from=c_from, transpose_flip=flip, variant=variant).double().numpy()
    ref = _wino43_panel_reference(w, c_from, flip, variant)
    assert u.size == ref.size + 64 and (u[ref.size:] == 0).all()
    assert np.abs(u[:ref.size] - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max())      # fp32 rounding of G g G^T


@pytest.mark.parametrize("what", ["wino_stream", "wino_a128", "wino43", "wino43_v1", "wino43_v2", "up2x_wino", "up2x_wino43", "dgrad_up2x_wino", "dgrad_up2x_wino43"])
def test_lds_dma_landing_as_late_as_the_waits_allow(emu, monkeypatch, what):
    """The emulator's LDS-DMA normally lands at issue -- as early as possible.  TNV3_EMU_LAZY_DMA=1 is the other extreme: a piece lands
    only when its work-item's counted s_waitcnt (or the kernel's end) forces it, and __syncthreads() forces nothing (hipcc emits no
    vmcnt for it).  A wait that counts one operation too many, or a barrier trusted to publish a DMA, then reads stale LDS here.
    (Mutation check when this was added: vmcnt(9) -> vmcnt(12) in the F(4x4) chunk loop passes the eager mode and fails this one.)"""
    from tracknetv3_amd import tuning
    monkeypatch.setenv("TNV3_EMU_LAZY_DMA", "1")
    monkeypatch.setenv("TNV3_EMU_CUS", "2")              # persistent workgroups walk several tiles: the pipelines cross tile boundaries
    if what == "wino_stream":
        monkeypatch.setattr(tuning, "WINO_VARIANT", 5)
        e_plain, e_full = _wino_case(1, 70, 64, 4, 128, "cpu")
        assert e_plain <= 3e-6 and e_full <= 6e-6
    elif what == "wino_a128":
        monkeypatch.setattr(tuning, "WINO_VARIANT", 6)
        e_plain, e_full = _wino_case(2, 12, 128, 8, 64, "cpu")
        assert e_plain <= 3e-6 and e_full <= 6e-6
    elif what in ("wino43", "wino43_v1", "wino43_v2"):
        v = {"wino43": 0, "wino43_v1": 1, "wino43_v2": 2}[what]
        _wino43_case((2, 20, 128, 16, 64), "cpu", variant=v)
        _wino43_case((1, 27, 64, 8, 128), "cpu", variant=v)
        _wino43_case((3, 9, 64, 16, 128), "cpu", variant=v)
    elif what == "up2x_wino":
        e_ref, e_old = _up2x_wino_case(2, 20, 128, 4, 64, "cpu")
        assert e_ref <= 3e-6 and e_old <= 4e-6
    elif what == "up2x_wino43":
        for case in ((2, 20, 128, 4, 64), (3, 40, 64, 6, 64)):
            e_ref, e_old = _up2x_wino_case(*case, "cpu", variant=2)
            assert e_ref <= 1.5e-5 and e_old <= 1.5e-5
    elif what == "dgrad_up2x_wino43":
        for case in ((2, 128, 24, 4, 32), (3, 64, 20, 6, 64)):
            e_ref, e_old = _dgrad_up2x_wino_case(*case, "cpu", variant=2)
            assert e_ref <= 1e-5 and e_old <= 1e-5
    else:
        e_ref, e_old = _dgrad_up2x_wino_case(*DGRAD_UP2X_WINO_CASES[1], "cpu")
        assert e_ref <= 3e-6 and e_old <= 4e-6


def test_wino_helpers_refuse_the_unresolved_variant(emu):
    """tnv3_conv3x3_wino_{stats_tiles, layout, has_stats}(-1) used to fall back to kernel 5 while the launchers resolve -1 through
    tnv3_conv3x3_wino_pick (kernel 6 for Cout % 128 == 0): a C caller following the header got half the statistics buffer / the wrong panel
    layout.  The helpers now fail on -1; resolved variants answer as before."""
    lib = emu.load()
    assert lib.tnv3_conv3x3_wino_layout(-1) < 0 and b"wino_pick" in lib.tnv3_last_error()
    assert lib.tnv3_conv3x3_wino_has_stats(-1) < 0
    assert lib.tnv3_conv3x3_wino_stats_tiles(2, 8, 64, -1) == 0
    v = lib.tnv3_conv3x3_wino_pick(64, 128)
    assert v == 6 and lib.tnv3_conv3x3_wino_layout(v) == 2 and lib.tnv3_conv3x3_wino_has_stats(v) == 1
    assert lib.tnv3_conv3x3_wino_stats_tiles(2, 8, 64, v) == 2 * 2 * 2 and lib.tnv3_conv3x3_wino_stats_tiles(2, 8, 64, 5) == 2 * 2 * 1


def test_wino43_takes_bn_constants_at_any_4_byte_offset(emu):
    """mean / scale / shift of the eval forward may be views at odd offsets (a flattened parameter buffer, a sliced state tensor):
    the 16x16x4 kernel reads them as per-lane scalars, only 4-byte alignment is needed.  Same bits as with aligned copies."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w = 1, 16, 64, 8, 64
    x, wt = torch.relu(T((n, cin, h, w), 491)), T((cout, cin, 3, 3), 492, -0.3, 0.3)
    flat = T((3 * cout + 8,), 493, 0.5, 1.5)
    mean, scale, shift = flat[1:1 + cout], flat[2 + cout:2 + 2 * cout], flat[5 + 2 * cout:5 + 3 * cout]
    assert mean.data_ptr() % 16 and scale.data_ptr() % 16 and shift.data_ptr() % 16
    u = ops.pack_wino43_weights(wt)
    got = ops.conv3x3_wino43(x, u, cout, mean=mean, scale=scale, shift=shift, relu=True)
    want = ops.conv3x3_wino43(x, u, cout, mean=mean.clone(), scale=scale.clone(), shift=shift.clone(), relu=True)
    assert torch.equal(got, want)
