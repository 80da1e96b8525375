"""CPU: training kernels (BN train fwd/bwd, dgrad, MFMA wgrad, WBCE, head backward, pool/upsample backward, mixup)
through the emulator, against torch autograd on the fp64 oracle restatement."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import nets, prng


def T(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(shape, seed, lo, hi))


def rel_err(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


def test_bn_train_forward_backward_emulated(emu):
    from tracknetv3_amd import ops, _lib
    n, c, h, w = 3, 70, 4, 8
    z = T((n, c, h, w), 1, -2, 3)
    g, b = T((c,), 2, 0.5, 1.5), T((c,), 3)
    rm, rv = T((c,), 4), T((c,), 5, 0.5, 2.0)
    rm0, rv0 = rm.clone(), rv.clone()
    a, mean, invstd = ops.bn_train_forward(z, g, b, rm, rv)
    zd = z.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    sd = {"bn.weight": gd, "bn.bias": bd, "bn.running_mean": rm0, "bn.running_var": rv0,
          "bn.num_batches_tracked": torch.tensor(0)}
    st = {}
    ref = torch.relu(nets.batchnorm2d(zd, sd, "bn", True, st))
    assert (a.double() - ref).abs().max().item() <= 2e-6
    assert torch.allclose(rm.double(), st["bn.running_mean"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(rv.double(), st["bn.running_var"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(mean.double(), zd.mean(dim=(0, 2, 3)), atol=1e-6)
    da = T((n, c, h, w), 6)
    ref.backward(da.double())
    dz, dgamma, dbeta = ops.bn_relu_backward(da.clone(), a, z, g, mean, invstd)
    assert rel_err(dz, zd.grad) <= 1e-5 and rel_err(dgamma, gd.grad) <= 1e-5 and rel_err(dbeta, bd.grad) <= 1e-5
    # mask recomputed from z instead of read from a: bit-identical results
    dz2, dgamma2, dbeta2 = ops.bn_relu_backward(da.clone(), None, z, g, mean, invstd, beta=b)
    assert torch.equal(dz2, dz) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)
    with pytest.raises(_lib.Tnv3Error):
        ops.bn_relu_backward(da.clone(), None, z, g, mean, invstd)


@pytest.mark.parametrize("case", [(2, 5, 0, 64, 6, 40, False), (1, 32, 16, 128, 4, 16, True), (2, 64, 0, 192, 4, 8, False),
                                  (1, 27, 0, 64, 5, 36, False)],
                         ids=["b_5to64", "a_dual_up_48to128", "b_64to192", "b_27to64_ragged"])
@pytest.mark.parametrize("variant", [0, 1], ids=["regstaged", "ldsdma"])
def test_wgrad_mfma_and_dgrad_emulated(monkeypatch, emu, case, variant):
    from tracknetv3_amd import ops
    from tracknetv3_amd import tuning
    monkeypatch.setattr(tuning, "WGRAD_VARIANT", variant)      # per-call kernel variant (the C ABI has no process-wide knob)
    _wgrad_case(case, ops)


def _wgrad_case(case, ops):
    n, c0, c1, cout, h, w, up = case
    s0 = T((n, c0, h // 2, w // 2) if up else (n, c0, h, w), 11)
    s1 = T((n, c1, h, w), 12) if c1 else None
    wt = T((cout, c0 + c1, 3, 3), 13, -0.3, 0.3)
    dz = T((n, cout, h, w), 14)
    x = s0.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else s0
    if c1:
        x = torch.cat([x, s1], 1)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    F.conv2d(xd, wd, padding=1).backward(dz.double())
    dw = ops.conv3x3_wgrad(s0, dz, src1=s1, up0=up)
    assert rel_err(dw, wd.grad) <= 2e-6
    if (c0 + c1) % 64 == 0:
        dx0, dx1 = ops.conv3x3_dgrad(dz, ops.pack_conv3x3_weights(wt, transpose_flip=True), c0, c1)
        assert rel_err(dx0, xd.grad[:, :c0]) <= 2e-6
        if c1:
            assert rel_err(dx1, xd.grad[:, c0:]) <= 2e-6


WGRAD_UP2X_CASES = [(1, 8, 16, 64, 4, 20), (2, 32, 32, 128, 3, 36), (1, 64, 32, 64, 6, 32),    # (n, c0, c1, cout, h_low, w_low)
                    (1, 64, 128, 128, 2, 16),                                              # skip half on the Winograd-form kernel
                    (1, 128, 64, 64, 4, 16), (2, 128, 16, 128, 3, 24), (1, 256, 64, 64, 2, 8),   # c0 % 128 == 0: upsampled half in the 9-GEMM form
                    (2, 40, 16, 64, 4, 24), (1, 33, 64, 128, 6, 8), (3, 8, 16, 64, 2, 40)]        # partial blocks of 32 input channels in the 25-of-36 form


def _wgrad_up2x_case(case, device):
    from tracknetv3_amd import ops
    n, c0, c1, cout, hl, wl = case
    xl, skip = T((n, c0, hl, wl), 31), T((n, c1, 2 * hl, 2 * wl), 32)
    dz = T((n, cout, 2 * hl, 2 * wl), 33)
    wd = T((cout, c0 + c1, 3, 3), 34, -0.3, 0.3).double().requires_grad_(True)
    x = torch.cat([xl.repeat_interleave(2, 2).repeat_interleave(2, 3), skip], 1)
    F.conv2d(x.double(), wd, padding=1).backward(dz.double())
    xl, skip, dz = xl.to(device), skip.to(device), dz.to(device)
    dw = ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=2, up_variant=1)      # 9-GEMM form (where c0 % 128 == 0) + F(2x2) kernel 1 for the skip half
    dw2 = ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=2, up_variant=1)
    assert torch.equal(dw, dw2), "split-K reduction must be deterministic"
    dwv = ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=1)                    # ABI 5's meaning of 1: the upsampled half by the four 2x2-window launches
    assert torch.equal(dwv, ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=2, up_variant=0))
    assert rel_err(dwv.cpu(), dw.cpu().double()) <= 4e-6
    # 9-GEMM form + the other (bit-identical) F(2x2) generation of the skip half's kernel
    assert torch.equal(dw, ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=5, up_variant=1))
    # the skip half by the F(4x4) kernel where it applies (H % 4 == 0, C1 % 64 == 0) -- same upsampled half, deterministic
    dwd = ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=-1, up_variant=1)
    assert torch.equal(dwd, ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=8, up_variant=1))
    assert torch.equal(dwd[:, :c0], dw[:, :c0])
    if (2 * hl) % 4 or c1 % 64:
        assert torch.equal(dwd, dw)
    assert rel_err(dwd.cpu(), wd.grad) <= 8e-6
    # the default: the upsampled half in the 25-of-36 F(4x4) form where the shape allows it (Hl % 2 == 0, Wl % 8 == 0, any c0) -- same skip half
    dwf = ops.conv3x3_wgrad_up2x(xl, skip, dz)
    assert torch.equal(dwf, ops.conv3x3_wgrad_up2x(xl, skip, dz, wino_variant=-1, up_variant=2)), "deterministic"
    assert torch.equal(dwf[:, c0:], dwd[:, c0:])
    if hl % 2 or wl % 8:
        assert torch.equal(dwf, dwd)
    # (the 25-of-36 form: 4-8e-6 of max|dW| from fp64 where the 9-GEMM F(2x2) form has 5e-7 -- a leaf gradient, nothing amplifies it)
    assert rel_err(dwf.cpu(), wd.grad) <= 1.5e-5 and rel_err(dwf.cpu()[:, :c0], wd.grad[:, :c0]) <= 1.5e-5, (rel_err(dwf.cpu(), wd.grad), rel_err(dwf.cpu()[:, :c0], wd.grad[:, :c0]))
    return rel_err(dw.cpu(), wd.grad), rel_err(dw.cpu()[:, :c0], wd.grad[:, :c0])


@pytest.mark.parametrize("case", WGRAD_UP2X_CASES)
def test_wgrad_up2x_emulated_vs_autograd(emu, case):
    e_all, e_up = _wgrad_up2x_case(case, "cpu")
    assert e_all <= 3e-6 and e_up <= 3e-6, (e_all, e_up)


WGRAD_WINO_CASES = [(1, 64, 64, 4, 16), (2, 64, 128, 6, 32), (1, 128, 64, 2, 48), (2, 27, 64, 6, 32), (1, 100, 64, 4, 16)]   # (n, cin, cout, h, w); 27: the stem


def _wgrad_wino_case(case, device):
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    x, dz = torch.relu(T((n, cin, h, w), 51)), T((n, cout, h, w), 52)
    wd = T((cout, cin, 3, 3), 53, -0.3, 0.3).double().requires_grad_(True)
    F.conv2d(x.double(), wd, padding=1).backward(dz.double())
    f22 = 1 if cin % 64 == 0 else 5
    dw = ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=f22)
    assert torch.equal(dw, ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=f22)), "split-K reduction must be deterministic"
    # the default (-1): the F(4x4) kernel where it applies (H % 4 == 0), else the F(2x2) generation above
    dwd = ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=-1)
    assert torch.equal(dwd, ops.conv3x3_wgrad_wino(x.to(device), dz.to(device)))
    assert torch.equal(dwd, ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=8) if h % 4 == 0 else dw)
    # every F(2x2) kernel generation accumulates every element in the same order: bit-identical gradients
    from tracknetv3_amd import _lib
    twins = _lib.is_emulator()                              # the emulator is built with -DTNV3_DIAG and dispatches the measurement twins too
    for v in (((1, 2, 5) if twins else (1, 5)) if cin % 64 == 0 else (5,)):        # a partial block of input channels (the stem): kernel 5 only
        assert torch.equal(dw, ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=v)), v
    for v in (() if twins else (0, 2, 3, 4, 6, 7)):         # measurement twins of libtnv3_diag.so since ABI 5 / 6: the product library refuses them
        with pytest.raises(Exception, match="libtnv3_diag"):
            ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=v)
    if cin % 64:
        with pytest.raises(Exception, match="Cin % 64"):
            ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=1)
    return rel_err(dw.cpu(), wd.grad)


# F(4x4) weight gradient (kernel variant 8; H % 4 == 0): not bit-identical to the F(2x2) generations -- against fp64 autograd.
# (n, cin, cout, h, w): 27 = the stem (a partial block of 32 input channels), 100 = three full blocks + a partial one; (3, 64, 64, 8, 48):
# strips over images, tile rows and columns; borders on every side
WGRAD_WINO43_CASES = [(1, 64, 64, 4, 16), (2, 64, 128, 8, 32), (2, 27, 64, 8, 32), (1, 100, 64, 4, 16), (3, 64, 64, 8, 48), (1, 32, 64, 12, 64)]


def _wgrad_wino43_case(case, device):
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    x, dz = torch.relu(T((n, cin, h, w), 51)), T((n, cout, h, w), 52)
    wd = T((cout, cin, 3, 3), 53, -0.3, 0.3).double().requires_grad_(True)
    F.conv2d(x.double(), wd, padding=1).backward(dz.double())
    dw = ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=8)
    assert torch.equal(dw, ops.conv3x3_wgrad_wino(x.to(device), dz.to(device), variant=8)), "split-K reduction must be deterministic"
    return rel_err(dw.cpu(), wd.grad)


@pytest.mark.parametrize("case", WGRAD_WINO43_CASES)
def test_wgrad_wino43_emulated_vs_autograd(emu, case):
    assert _wgrad_wino43_case(case, "cpu") <= 8e-6


@pytest.mark.parametrize("cus", [1, 2, 3, 5])
def test_wgrad_wino43_long_strip_walks_emulated(emu, monkeypatch, cus):
    """Few CUs -> small split-K -> every workgroup walks many strips (column, tile-row and image carries of the cursors; loads two steps
    ahead; shares of 1, 2 strips and uneven shares)."""
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    assert _wgrad_wino43_case((3, 64, 64, 8, 48), "cpu") <= 8e-6
    assert _wgrad_wino43_case((2, 27, 64, 4, 32), "cpu") <= 8e-6


def test_wgrad_wino43_with_lds_dma_landing_late(emu, monkeypatch):
    monkeypatch.setenv("TNV3_EMU_LAZY_DMA", "1")
    monkeypatch.setenv("TNV3_EMU_CUS", "2")
    assert _wgrad_wino43_case((2, 27, 64, 8, 32), "cpu") <= 8e-6


@pytest.mark.parametrize("lazy", ["0", "1"])
@pytest.mark.parametrize("sw", [32, 512, 256, 256 + 32, 256 + 512])
def test_wgrad_wino43_candidate_schedules_are_bit_identical(emu, monkeypatch, sw, lazy):
    """The schedule switches of wgrad_wino43_body (WgradWino43Sw: the dY loads issued behind quad 3 / quad 1 instead of at the end of a step,
    three raw stages with the DMA a step further ahead) change WHEN operands are requested, never what is summed in which order:
    bit-identical to the product kernel, with the LDS-DMA landing at issue and as late as the waits allow, strips over images / tile rows /
    columns, a partial block of input channels, long walks (few CUs)."""
    from tracknetv3_amd import ops
    monkeypatch.setenv("TNV3_EMU_LAZY_DMA", lazy)
    for cus, (n, cin, cout, h, w) in ((2, (3, 64, 64, 8, 48)), (3, (2, 27, 64, 8, 32)), (256, (2, 64, 128, 8, 32)), (1, (1, 100, 64, 4, 16))):
        monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
        x, dz = torch.relu(T((n, cin, h, w), 51)), T((n, cout, h, w), 52)
        assert torch.equal(ops.conv3x3_wgrad_wino(x, dz, variant=8), ops.conv3x3_wgrad_wino(x, dz, variant=8000 + sw)), (sw, cus)


def test_wgrad_wino43_refuses_heights_that_are_not_a_multiple_of_four(emu):
    from tracknetv3_amd import ops
    with pytest.raises(Exception, match="H % 4"):
        ops.conv3x3_wgrad_wino(T((1, 64, 6, 16), 1), T((1, 64, 6, 16), 2), variant=8)


# emulator-only: 5 and 7 strips of work -> split-K counts whose quarters are uneven / partly empty in the fold kernel
# (3, 64, 64, 4, 48): 18 strips over images, tile rows and segments -- the chunk walk's carries; borders on every side
WGRAD_WINO_EMU_EXTRA = [(1, 64, 64, 10, 16), (1, 64, 64, 14, 16), (3, 64, 64, 4, 48)]


@pytest.mark.parametrize("case", WGRAD_WINO_CASES + WGRAD_WINO_EMU_EXTRA)
def test_wgrad_wino_emulated_vs_autograd(emu, case):
    assert _wgrad_wino_case(case, "cpu") <= 4e-6


@pytest.mark.parametrize("cus", [1, 2, 3, 5])
def test_wgrad_up2x_wino_long_chunk_walks_emulated(emu, monkeypatch, cus):
    """The 9-GEMM weight gradient of the upsampled half with few CUs: every workgroup walks many 8-pixel strips (cursor carries
    over segments, rows and images; borders on every side)."""
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    e_all, e_up = _wgrad_up2x_case((3, 128, 16, 64, 4, 24), "cpu")
    assert e_all <= 3e-6 and e_up <= 3e-6, (e_all, e_up)


@pytest.mark.parametrize("cus", [1, 2, 3, 5])
def test_wgrad_wino_long_chunk_walks_emulated(emu, monkeypatch, cus):
    """Few CUs -> small split-K -> every workgroup walks many strips (steps of 1, 2, 3, 5 strips: segment, tile-row and image
    carries of the division-free cursor, DMA two chunks ahead, both wave groups' phase loops)."""
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    assert _wgrad_wino_case((3, 64, 64, 4, 48), "cpu") <= 4e-6
    assert _wgrad_wino_case((2, 64, 128, 6, 32), "cpu") <= 4e-6


def test_dgrad_split_destinations_emulated(emu):
    from tracknetv3_amd import ops
    n, c0, c1, cout, h, w = 1, 128, 64, 64, 4, 8          # shapes of up_block_3.conv_1's data gradient
    wt, dz = T((cout, c0 + c1, 3, 3), 21, -0.3, 0.3), T((n, cout, h, w), 22)
    xd = torch.zeros((n, c0 + c1, h, w), dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, wt.double(), padding=1).backward(dz.double())
    dx0, dx1 = ops.conv3x3_dgrad(dz, ops.pack_conv3x3_weights(wt, transpose_flip=True), c0, c1)
    assert dx0.shape == (n, c0, h, w) and dx1.shape == (n, c1, h, w)
    assert rel_err(dx0, xd.grad[:, :c0]) <= 2e-6 and rel_err(dx1, xd.grad[:, c0:]) <= 2e-6


def test_wbce_forward_backward_emulated(emu):
    from tracknetv3_amd.utils.metric import WBCELoss
    g = np.load(os.path.join(GOLDEN, "wbce_edge.npz"))
    p = torch.from_numpy(g["p"]).requires_grad_(True)
    y = torch.from_numpy(g["y"])
    loss = WBCELoss(p, y)
    assert loss.shape == () and abs(loss.item() - float(g["loss"])) <= 2e-6
    loss.backward()
    np.testing.assert_allclose(p.grad.numpy(), g["grad"], rtol=2e-5, atol=1e-8)
    per = WBCELoss(p.detach(), y, reduce=False)
    assert per.shape == (1,) and abs(per[0].item() - float(g["per_sample"][0])) <= 2e-6
    # random maps, per-sample reduction with a non-trivial upstream gradient
    pp = T((3, 2, 8, 16), 5, 0.01, 0.99).requires_grad_(True)
    yy = (T((3, 2, 8, 16), 6, 0, 1) > 0.9).float()
    up = torch.tensor([0.5, -1.0, 2.0])
    (WBCELoss(pp, yy, reduce=False) * up).sum().backward()
    pd = pp.detach().double().requires_grad_(True)
    (nets.wbce_loss(pd, yy.double(), reduce=False) * up.double()).sum().backward()
    assert rel_err(pp.grad, pd.grad) <= 1e-5


def test_head_pool_upsample_mixup_backward_emulated(emu):
    from tracknetv3_amd import ops
    n, L, h, w = 2, 3, 6, 40            # HW = 240: one full 128-pixel tile + a ragged one per sample
    a = T((n, 64, h, w), 1)
    wt, b = T((L, 64, 1, 1), 2, -0.3, 0.3), T((L,), 3)
    ad, wd, bd = a.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    pref = torch.sigmoid(F.conv2d(ad, wd, bd))
    dp = T((n, L, h, w), 4)
    pref.backward(dp.double())
    p = ops.head1x1_sigmoid(a, wt, b)
    da, dw, db = ops.head_backward(dp, p, a, wt)
    assert rel_err(da, ad.grad) <= 1e-5 and rel_err(dw, wd.grad) <= 1e-5 and rel_err(db, bd.grad) <= 1e-5
    for (n2, L2, h2, w2) in ((1, 16, 5, 7), (3, 1, 4, 33)):   # H*W not a multiple of 4 (element-wise loads), L = 16 and L = 1
        a2, wt2, b2 = T((n2, 64, h2, w2), 11), T((L2, 64, 1, 1), 12, -0.3, 0.3), T((L2,), 13)
        ad2, wd2, bd2 = a2.double().requires_grad_(True), wt2.double().requires_grad_(True), b2.double().requires_grad_(True)
        pref2 = torch.sigmoid(F.conv2d(ad2, wd2, bd2))
        dp2 = T((n2, L2, h2, w2), 14)
        pref2.backward(dp2.double())
        da2, dw2, db2 = ops.head_backward(dp2, pref2.detach().float().contiguous(), a2, wt2)   # (the forward head needs H*W % 4 == 0)
        assert rel_err(da2, ad2.grad) <= 1e-5 and rel_err(dw2, wd2.grad) <= 1e-5 and rel_err(db2, bd2.grad) <= 1e-5
    # max-pool backward (+ skip add), incl. ties at zero after a ReLU
    x = torch.relu(T((2, 3, 8, 12), 5))
    xd = x.double().requires_grad_(True)
    dpool, dskip = T((2, 3, 4, 6), 6), T((2, 3, 8, 12), 7)
    F.max_pool2d(xd, 2, 2).backward(dpool.double())
    got = ops.maxpool2x2_backward_add(x, dpool, dskip)
    assert (got.double() - (xd.grad + dskip.double())).abs().max().item() <= 1e-6
    # nearest-upsample backward
    dhi = T((2, 3, 8, 12), 8)
    lo = torch.zeros((2, 3, 4, 6), dtype=torch.float64, requires_grad=True)
    nets.upsample2x_nearest(lo).backward(dhi.double())
    assert (ops.upsample2x_backward(dhi).double() - lo.grad).abs().max().item() <= 1e-6
    # mixup with injected lambda / permutation (train.py:32-40)
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    x = nets.synth_input((4, 3, 8, 16), 11)
    lam = np.maximum(g["mixup_lam"], 1 - g["mixup_lam"]).astype(np.float32)
    out = ops.mixup(x, torch.from_numpy(lam), torch.from_numpy(g["mixup_perm"].astype(np.int32)))
    assert np.abs(out.numpy() - g["mixup_x"]).max() <= 1e-6


def test_head_backward_many_tiles_per_workgroup_emulated(emu):
    """More pixel tiles than workgroups (1025 x 2 > 1024): every workgroup walks several tiles, so the register prefetch of the
    next tile overlaps the arithmetic of the current one -- the path every real training step takes (11 520 tiles)."""
    from tracknetv3_amd import ops
    n, L, h, w = 2, 2, 1025, 128
    a = T((n, 64, h, w), 31)
    wt, b = T((L, 64, 1, 1), 32, -0.3, 0.3), T((L,), 33)
    ad, wd, bd = a.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    pref = torch.sigmoid(F.conv2d(ad, wd, bd))
    dp = T((n, L, h, w), 34)
    pref.backward(dp.double())
    da, dw, db = ops.head_backward(dp, pref.detach().float().contiguous(), a, wt)
    assert rel_err(da, ad.grad) <= 1e-5 and rel_err(dw, wd.grad) <= 1e-5 and rel_err(db, bd.grad) <= 1e-5


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("TNV3_EMU_FULL") != "1", reason="minutes of emulation; set TNV3_EMU_FULL=1 (the GPU suite runs the same check)")
def test_tracknet_train_step_emulated_vs_fp64_oracle(emu):
    """forward(train) + WBCELoss + backward of the whole TrackNet(9,3) through the product's autograd node."""
    from tracknetv3_amd.model import TrackNet
    from tracknetv3_amd.utils.metric import WBCELoss
    in_dim, out_dim, seed = 9, 3, 13
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, out_dim), seed, calibrated=True)
    m = TrackNet(in_dim, out_dim)
    m.load_state_dict(sd, strict=True)
    m.train()
    x = nets.synth_input((2, in_dim, 8, 32), seed + 1000)
    y = nets.disc_heatmaps(2, out_dim, 8, 32, seed + 2000)
    p = m(x)
    loss = WBCELoss(p, y)
    loss.backward()
    l64, p64, g64, st64 = nets.tracknet_train_step_grads(sd, x, y, torch.float64)
    assert abs(loss.item() - l64.item()) <= 1e-5
    assert (p.detach().double() - p64).abs().max().item() <= 5e-5
    # gradient tolerance: the fp32 reference arithmetic itself deviates from fp64 by ~1e-2 of max|g| (BN
    # cancellation); require our error to be of that order, parameter by parameter
    _, _, g32, _ = nets.tracknet_train_step_grads(sd, x, y, torch.float32)
    mine = np.array([rel_err(prm.grad, g64[name]) for name, prm in m.named_parameters()])
    ref = np.array([rel_err(g32[name], g64[name]) for name, _ in m.named_parameters()])
    assert mine.max() <= 3 * ref.max() + 2e-4, (mine.max(), ref.max())
    assert np.median(mine) <= 3 * np.median(ref) + 1e-4
    after = m.state_dict()
    for k, v in st64.items():
        if "num_batches" in k:
            assert int(after[k]) == 1
        else:
            assert torch.allclose(after[k].double(), v, rtol=1e-4, atol=1e-6), k


def test_train_step_releases_its_activations_without_the_cyclic_gc(emu):
    """The autograd nodes must not form reference cycles with their outputs: with the cyclic collector disabled, the
    network output (and with it every saved activation of the step) has to die with the last user reference.
    (InpaintNet here; the TrackNet node is checked on the GPU, a whole emulated TrackNet step takes minutes.)"""
    import gc
    import weakref
    from tracknetv3_amd.model import InpaintNet
    gc.collect()
    gc.disable()
    try:
        net = InpaintNet().train()
        out = net(nets.synth_input((2, 16, 2), 7), (nets.synth_input((2, 16, 1), 8) > 0.5).float())
        out.sum().backward()
        alive = weakref.ref(out)
        del out
        assert alive() is None, "InpaintNet output survived: ctx <-> output reference cycle"
    finally:
        gc.enable()


def test_inpaintnet_train_step_emulated_vs_reference_golden(emu):
    """InpaintNet forward(train) + masked MSE (train.py:159-161) + backward through the product's autograd node."""
    from tracknetv3_amd.model import InpaintNet
    g = np.load(os.path.join(GOLDEN, "inpaintnet_6x16.npz"))
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 77)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net.train()
    n, L = 6, 16
    coor = nets.synth_input((n, L, 2), 501)
    vis = (nets.synth_input((n, L, 1), 502) > 0.2).float()
    coor = coor * vis
    mask = ((nets.synth_input((n, L, 1), 503) < 0.3).float() * vis)
    gt = nets.synth_input((n, L, 2), 504)
    out = net(coor * (1 - mask), mask)
    loss = torch.nn.MSELoss()(out * mask, gt * mask)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-7
    assert np.abs(out.detach().numpy() - g["out"]).max() <= 2e-6
    params = dict(net.named_parameters())
    for k, name in enumerate(g["grad_names"]):
        gr = params[str(name)].grad.double()
        assert abs(gr.sum().item() - g["grad_sums"][k]) <= 2e-4 * g["grad_abs"][k] + 1e-9, name
        assert abs(gr.abs().sum().item() - g["grad_abs"][k]) <= 2e-4 * g["grad_abs"][k] + 1e-9, name
    assert rel_err(params["predictor.weight"].grad, torch.from_numpy(g["grad_pred_w"])) <= 2e-5
    assert rel_err(params["down_1.conv.weight"].grad, torch.from_numpy(g["grad_down1_w"])) <= 2e-4
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1)          # train.py:165 works on the leaf parameters


BN_BWD_EPILOGUE_CASES = [(2, 16, 128, 8, 64), (1, 24, 64, 8, 128), (2, 64, 64, 4, 64)]


def _bn_bwd_epilogue_case(case, device):
    """conv3x3_wino_dgrad_bnstats + bn_relu_backward_tiles == conv3x3_wino + bn_relu_backward: the same dA bits, the same dZ /
    dgamma / dbeta up to the fp64 summation order of the two sums (variant 6 for 128 output channels, variant 5 for 64)."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case            # the data gradient maps cin = the next block's Cout -> cout = this block's channels
    dz_next, wt = T((n, cin, h, w), 701).to(device), T((cin, cout, 3, 3), 702, -0.3, 0.3).to(device)   # weight of the NEXT block: [Cout_next = cin][cout]
    z = T((n, cout, h, w), 703, -1.0, 1.0).to(device)
    gamma, beta = T((cout,), 704, 0.5, 1.5).to(device), T((cout,), 705, -0.3, 0.3).to(device)
    mean = z.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.var((0, 2, 3), unbiased=False) + 1e-5)
    u_t = ops.pack_wino_weights(wt, transpose_flip=True)
    da_ref = ops.conv3x3_wino(dz_next, u_t, cout)
    dz_ref, dg_ref, db_ref = ops.bn_relu_backward(da_ref.clone(), None, z, gamma, mean, invstd, beta=beta)
    c4 = ops.bn_bwd_consts(mean, invstd, gamma, beta)
    da, st = ops.conv3x3_wino_dgrad_bnstats(dz_next, u_t, cout, z, c4)
    assert torch.equal(da, da_ref)
    # the tile sums ARE dbeta / dgamma (same fp32 mask expression, same fp64 accumulation, another summation order)
    scale = max(da.abs().sum((0, 2, 3)).max().item(), 1.0)
    assert (st.sum(1)[:, 0] - db_ref.double()).abs().max().item() <= 2e-6 * scale
    assert (st.sum(1)[:, 1] - dg_ref.double()).abs().max().item() <= 2e-6 * scale
    dz2, dg2, db2 = ops.bn_relu_backward_tiles(da, z, gamma, beta, mean, invstd, st)
    assert rel_err(dg2, dg_ref) <= 1e-6 and rel_err(db2, db_ref) <= 1e-6 and rel_err(dz2, dz_ref) <= 1e-6


@pytest.mark.parametrize("case", BN_BWD_EPILOGUE_CASES)
def test_bn_backward_sums_from_the_data_gradient_epilogue_emulated(emu, monkeypatch, case):
    monkeypatch.setenv("TNV3_EMU_CUS", "8")
    _bn_bwd_epilogue_case(case, "cpu")


# (n, cin = the next block's Cout, cout = this block's channels, h, w): 128 -> the 128-channel geometry, 64 -> the 64-channel one (two tile
# rows per workgroup; h = 12: a half-empty last tile row), several tiles per workgroup
BN_BWD_EPILOGUE43_CASES = [(2, 16, 128, 8, 64), (1, 32, 64, 12, 128), (2, 64, 64, 4, 64)]


def _bn_bwd_epilogue43_case(case, device):
    """VERDICT r5 #3 -- the F(4x4) twin: conv3x3_wino43_dgrad_bnstats + bn_relu_backward_tiles == conv3x3_wino43 + bn_relu_backward: the same dA
    bits, the same mask expression, the same fp64 accumulation of the two sums in another order (per 4 x 64 pixel tile, then the tiles)."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w = case
    dz_next, wt = T((n, cin, h, w), 711).to(device), T((cin, cout, 3, 3), 712, -0.3, 0.3).to(device)   # weight of the NEXT block: [Cout_next = cin][cout]
    z = T((n, cout, h, w), 713, -1.0, 1.0).to(device)
    gamma, beta = T((cout,), 714, 0.5, 1.5).to(device), T((cout,), 715, -0.3, 0.3).to(device)
    mean = z.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.var((0, 2, 3), unbiased=False) + 1e-5)
    u_t = ops.pack_wino43_weights(wt, transpose_flip=True)
    da_ref = ops.conv3x3_wino43(dz_next, u_t, cout)
    dz_ref, dg_ref, db_ref = ops.bn_relu_backward(da_ref.clone(), None, z, gamma, mean, invstd, beta=beta)
    da, st = ops.conv3x3_wino43_dgrad_bnstats(dz_next, u_t, cout, z, mean, invstd, gamma, beta)
    assert torch.equal(da, da_ref)
    assert tuple(st.shape) == (cout, n * (h // 4) * (w // 64), 2)
    scale = max(da.abs().sum((0, 2, 3)).max().item(), 1.0)
    assert (st.sum(1)[:, 0] - db_ref.double()).abs().max().item() <= 2e-6 * scale
    assert (st.sum(1)[:, 1] - dg_ref.double()).abs().max().item() <= 2e-6 * scale
    dz2, dg2, db2 = ops.bn_relu_backward_tiles(da, z, gamma, beta, mean, invstd, st)
    assert rel_err(dg2, dg_ref) <= 1e-6 and rel_err(db2, db_ref) <= 1e-6 and rel_err(dz2, dz_ref) <= 1e-6
    da3, st3 = ops.conv3x3_wino43_dgrad_bnstats(dz_next, u_t, cout, z, mean, invstd, gamma, beta)      # deterministic
    assert torch.equal(da3, da_ref) and torch.equal(st3, st)                                           # (da itself was turned into dZ in place above)


@pytest.mark.parametrize("cus", [8, 2])
@pytest.mark.parametrize("case", BN_BWD_EPILOGUE43_CASES)
def test_bn_backward_sums_from_the_f43_data_gradient_epilogue_emulated(emu, monkeypatch, case, cus):
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    _bn_bwd_epilogue43_case(case, "cpu")


DGRAD_UP2X_BNSUMS_CASES = [(1, 128, 16, 2, 32), (2, 64, 24, 6, 32), (1, 192, 9, 2, 64)]


def _dgrad_up2x_bnsums_case(case, device):
    """dgrad_up2x_wino_bnstats + bn_relu_backward_tiles == dgrad_up2x_wino(variant 2) + bn_relu_backward: the same d_low bits, the same mask expression,
    the two sums accumulated in fp64 per tile row of 2 x 32 low-resolution pixels (both geometries; a half-empty last tile row)."""
    from tracknetv3_amd import ops
    n, c0, cout, hl, wl = case
    dz, wt = T((n, cout, 2 * hl, 2 * wl), 741).to(device), T((cout, c0 + 8, 3, 3), 742, -0.3, 0.3).to(device)
    z = T((n, c0, hl, wl), 743, -1.0, 1.0).to(device)
    gamma, beta = T((c0,), 744, 0.5, 1.5).to(device), T((c0,), 745, -0.3, 0.3).to(device)
    mean = z.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(z.var((0, 2, 3), unbiased=False) + 1e-5)
    u = ops.pack_dgrad_up2x_wino_weights(wt, c0, variant=2)
    d_ref = ops.dgrad_up2x_wino(dz, u, c0, variant=2)
    dz_ref, dg_ref, db_ref = ops.bn_relu_backward(d_ref.clone(), None, z, gamma, mean, invstd, beta=beta)
    d, st = ops.dgrad_up2x_wino_bnstats(dz, u, c0, z, mean, invstd, gamma, beta)
    assert torch.equal(d, d_ref) and tuple(st.shape) == (c0, n * (hl // 2) * (wl // 32), 2)
    scale = max(d.abs().sum((0, 2, 3)).max().item(), 1.0)
    assert (st.sum(1)[:, 0] - db_ref.double()).abs().max().item() <= 2e-6 * scale
    assert (st.sum(1)[:, 1] - dg_ref.double()).abs().max().item() <= 2e-6 * scale
    dz2, dg2, db2 = ops.bn_relu_backward_tiles(d, z, gamma, beta, mean, invstd, st)
    assert rel_err(dg2, dg_ref) <= 1e-6 and rel_err(db2, db_ref) <= 1e-6 and rel_err(dz2, dz_ref) <= 1e-6
    d3, st3 = ops.dgrad_up2x_wino_bnstats(dz, u, c0, z, mean, invstd, gamma, beta)                # deterministic
    assert torch.equal(d3, d_ref) and torch.equal(st3, st)


@pytest.mark.parametrize("cus", [8, 2])
@pytest.mark.parametrize("case", DGRAD_UP2X_BNSUMS_CASES)
def test_bn_backward_sums_from_the_upsampled_half_data_gradient_emulated(emu, monkeypatch, case, cus):
    monkeypatch.setenv("TNV3_EMU_CUS", str(cus))
    _dgrad_up2x_bnsums_case(case, "cpu")


def _bn_apply_pool_case(case, device):
    """bn_train_forward(pool=True) == bn_train_forward + maxpool2x2, bit for bit (a, the saved statistics, the running statistics, the pooled tensor)."""
    from tracknetv3_amd import ops
    n, c, h, w = case
    z = T((n, c, h, w), 731, -1.0, 1.0).to(device)
    gamma, beta = T((c,), 732, 0.5, 1.5).to(device), T((c,), 733, -0.3, 0.3).to(device)
    st = torch.stack([z.double().sum((0, 2, 3)), (z.double() ** 2).sum((0, 2, 3))], 1).reshape(c, 1, 2).contiguous()
    rm0, rv0 = torch.zeros(c, device=device), torch.ones(c, device=device)
    rm1, rv1 = rm0.clone(), rv0.clone()
    a0, m0, i0 = ops.bn_train_forward(z, gamma, beta, rm0, rv0, 1e-5, 0.1, tile_stats=st)
    a1, m1, i1, p1 = ops.bn_train_forward(z, gamma, beta, rm1, rv1, 1e-5, 0.1, tile_stats=st, pool=True)
    assert torch.equal(a0, a1) and torch.equal(m0, m1) and torch.equal(i0, i1) and torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
    assert torch.equal(p1, ops.maxpool2x2(a0)) and torch.equal(p1.cpu(), F.max_pool2d(a0.cpu(), 2, 2))
    a2, _, _, p2 = ops.bn_train_forward(z, gamma, beta, rm1.clone(), rv1.clone(), 1e-5, 0.1, pool=True)      # (no tile statistics: the two-pass fallback)
    assert torch.equal(p2, ops.maxpool2x2(a2))


@pytest.mark.parametrize("case", [(2, 3, 8, 12), (1, 5, 2, 4), (3, 2, 6, 260)])
def test_bn_apply_that_also_writes_the_pooled_tensor_emulated(emu, case):
    _bn_apply_pool_case(case, "cpu")


POOL_BNSUMS_CASES = [(2, 3, 8, 12, True), (3, 5, 4, 8, False), (1, 2, 6, 260, True), (10, 4, 16, 64, True)]


def _pool_bnsums_case(case, device):
    """maxpool2x2_backward_add_bnstats + bn_relu_backward_tiles == bn_train_forward's a -> maxpool2x2_backward_add -> bn_relu_backward: the same dx
    bits (the activation is recomputed from z with the forward's expression: the same first maximum per window, ties at zero behind the ReLU
    included), the same mask, the two sums accumulated in fp64 in another order."""
    from tracknetv3_amd import ops
    n, c, h, w, skip = case
    z = T((n, c, h, w), 721, -1.0, 1.0).to(device)
    z[:, :, ::2, ::4] = z[:, :, 1::2, 1::4]                 # ties inside windows, above and below zero
    gamma, beta = T((c,), 722, 0.5, 1.5).to(device), T((c,), 723, -0.3, 0.3).to(device)
    rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
    a, mean, invstd = ops.bn_train_forward(z, gamma, beta, rm, rv, 1e-5, 0.1)
    dpool = T((n, c, h // 2, w // 2), 724).to(device)
    dskip = T((n, c, h, w), 725).to(device) if skip else None
    dx_ref = ops.maxpool2x2_backward_add(a, dpool, dskip)
    dz_ref, dg_ref, db_ref = ops.bn_relu_backward(dx_ref.clone(), None, z, gamma, mean, invstd, beta=beta)
    assert ops.maxpool2x2_bnstats_supported(n, h, w)
    dx, st = ops.maxpool2x2_backward_add_bnstats(z, dpool, dskip, mean, invstd, gamma, beta)
    assert torch.equal(dx, dx_ref) and st.shape[0] == c and st.shape[2] == 2
    scale = max(dx.abs().sum((0, 2, 3)).max().item(), 1.0)
    assert (st.sum(1)[:, 0] - db_ref.double()).abs().max().item() <= 2e-6 * scale
    assert (st.sum(1)[:, 1] - dg_ref.double()).abs().max().item() <= 2e-6 * scale
    dz2, dg2, db2 = ops.bn_relu_backward_tiles(dx, z, gamma, beta, mean, invstd, st)
    assert rel_err(dg2, dg_ref) <= 1e-6 and rel_err(db2, db_ref) <= 1e-6 and rel_err(dz2, dz_ref) <= 1e-6
    dx3, st3 = ops.maxpool2x2_backward_add_bnstats(z, dpool, dskip, mean, invstd, gamma, beta)          # deterministic
    assert torch.equal(dx3, dx_ref) and torch.equal(st3, st)
    assert not ops.maxpool2x2_bnstats_supported(n, h, w + 2)
    with pytest.raises(Exception):
        ops.maxpool2x2_backward_add_bnstats(z[:, :, :, :w - 2].contiguous(), dpool[:, :, :, :w // 2 - 1].contiguous(), None, mean, invstd, gamma, beta)


@pytest.mark.parametrize("case", POOL_BNSUMS_CASES)
def test_bn_backward_sums_from_the_max_pool_backward_emulated(emu, case):
    _pool_bnsums_case(case, "cpu")


def _inpaint_train_grads(net, coor, mask, gt):
    for p in net.parameters():
        p.grad = None
    out = net(coor * (1 - mask), mask)
    torch.nn.MSELoss()(out * mask, gt * mask).backward()
    return out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def test_inpaintnet_fused_training_kernels_vs_layer_kernels_and_fp64_autograd(emu, monkeypatch):
    """The three-launch training step (kernels/inpaint_fused_train.h: forward that saves activations, one fused data-gradient
    kernel, one launch for every dW / db) against the per-layer kernels and against fp64 autograd of the oracle, on a ragged batch
    with more sequences than workgroups (8 emulated CUs -> 16 workgroups, 19 sequences: some carry two)."""
    from tracknetv3_amd import inpaint_ops
    from tracknetv3_amd.model import InpaintNet
    monkeypatch.setenv("TNV3_EMU_CUS", "8")
    sd = nets.synth_state(nets.inpaintnet_state_shapes(), 79)
    net = InpaintNet()
    net.load_state_dict(sd, strict=True)
    net.train()
    n, L = 19, 16
    coor, gt = nets.synth_input((n, L, 2), 601), nets.synth_input((n, L, 2), 602)
    mask = (nets.synth_input((n, L, 1), 603) < 0.4).float()
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "1")
    out_f, g_f = _inpaint_train_grads(net, coor, mask, gt)
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "0")
    out_l, g_l = _inpaint_train_grads(net, coor, mask, gt)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    o64 = nets.inpaintnet_forward(sd64, (coor * (1 - mask)).double(), mask.double())
    (((o64 - gt.double()) * mask.double()) ** 2).mean().backward()
    assert (out_f - out_l).abs().max().item() <= 2e-6 and (out_f.double() - o64.detach()).abs().max().item() <= 2e-6
    for name in g_f:
        ref = sd64[name].grad
        s = ref.abs().max().item()
        assert (g_f[name].double() - ref).abs().max().item() <= 2e-5 * s + 1e-10, name
        assert (g_l[name].double() - ref).abs().max().item() <= 2e-5 * s + 1e-10, name
    # deterministic: a second fused step gives the same bits; and the transposed pack follows in-place weight updates
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "1")
    _, g_f2 = _inpaint_train_grads(net, coor, mask, gt)
    assert all(torch.equal(g_f[k], g_f2[k]) for k in g_f)
    with torch.no_grad():
        net.up_2.conv.weight.mul_(1.25)
        net.down_3.conv.bias.add_(0.05)
    _, g_new = _inpaint_train_grads(net, coor, mask, gt)
    monkeypatch.setattr(inpaint_ops, "FUSED_TRAIN", "0")
    _, g_new_l = _inpaint_train_grads(net, coor, mask, gt)
    for name in g_new:
        assert rel_err(g_new[name], g_new_l[name]) <= 2e-5, name


def test_head_sigmoid_wbce_fused_both_directions_emulated(emu):
    """sigmoid + WBCELoss fused into the head: the one-pass forward equals head -> WBCELoss, and the backward without a dP tensor
    equals wbce_backward -> head_backward (same element arithmetic)."""
    from tracknetv3_amd import ops
    n, L, h, w = 2, 3, 8, 40
    a = T((n, 64, h, w), 41)
    wt, b = T((L, 64, 1, 1), 42, -0.4, 0.4), T((L,), 43)
    y = (T((n, L, h, w), 44) > 0.6).float() * T((n, L, h, w), 45, 0.2, 1.0)          # fractional targets, as after mixup
    for reduce in (True, False):
        p_ref = ops.head1x1_sigmoid(a, wt, b)
        loss_ref = ops.wbce_forward(p_ref, y, reduce)
        p, loss = ops.head1x1_sigmoid_wbce(a, wt, b, y, reduce)
        assert torch.equal(p, p_ref)
        assert torch.allclose(loss, loss_ref, rtol=1e-6, atol=1e-9)
        up = T((1 if reduce else n,), 46, 0.5, 1.5)
        dp = ops.wbce_backward(p_ref, y, up, reduce)
        da_ref, dw_ref, db_ref = ops.head_backward(dp, p_ref, a, wt)
        da, dw, db = ops.head_wbce_backward(y, p, a, wt, up, reduce)
        assert rel_err(da, da_ref) <= 1e-6 and rel_err(dw, dw_ref) <= 1e-6 and rel_err(db, db_ref) <= 1e-6


@pytest.mark.parametrize("variant", [3, 5], ids=["balanced", "persistent"])
@pytest.mark.parametrize("case", [(2, 16, 64, 8, 64, False), (1, 24, 128, 4, 128, True), (3, 16, 64, 12, 128, True)], ids=["plain", "with_addend", "18_tiles"])
def test_bn_statistics_from_the_conv_epilogue_emulated(emu, monkeypatch, case, variant):
    """conv3x3_wino_stats: the raw output is bit-identical to conv3x3_wino and the per-tile (sum, sum of squares) fold to the
    batch statistics; bn_train_forward(tile_stats=...) equals the pass over z (same finalize arithmetic)."""
    from tracknetv3_amd import ops
    n, cin, cout, h, w, with_add = case
    x, wt = torch.relu(T((n, cin, h, w), 61)), T((cout, cin, 3, 3), 62, -0.3, 0.3)
    add = T((n, cout, h, w), 63) if with_add else None
    u = ops.pack_wino_weights(wt)
    z_ref = ops.conv3x3_wino(x, u, cout, addend=add, variant=3)
    monkeypatch.setenv("TNV3_EMU_CUS", "8")           # the persistent kernel walks several tiles per workgroup in the third case
    z, stats = ops.conv3x3_wino_stats(x, u, cout, addend=add, variant=variant)
    assert torch.equal(z, z_ref)
    assert stats.shape == (cout, n * (h // 4) * (w // 64), 2) and stats.dtype == torch.float64
    s1, s2 = stats[:, :, 0].sum(1), stats[:, :, 1].sum(1)
    zd = z.double()
    assert torch.allclose(s1, zd.sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9) and torch.allclose(s2, (zd * zd).sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    g, b = T((cout,), 64, 0.5, 1.5), T((cout,), 65)
    outs = []
    for ts in (None, stats):
        rm, rv = T((cout,), 66), T((cout,), 67, 0.5, 2.0)
        a, mean, invstd = ops.bn_train_forward(z, g, b, rm, rv, tile_stats=ts)
        outs.append((a, mean, invstd, rm, rv))
    for p_, q_ in zip(outs[0], outs[1]):
        assert torch.allclose(p_, q_, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", [(1, 64, 64, 4, 16), (2, 27, 64, 6, 32)])
def test_wgrad_wino_with_lds_dma_landing_late(emu, monkeypatch, case):
    """Every Winograd weight-gradient generation with the emulator's LDS-DMA landing as late as the counted waits allow
    (TNV3_EMU_LAZY_DMA=1, tests/emu/hip_emu.h): same bit-identical results, same error."""
    monkeypatch.setenv("TNV3_EMU_LAZY_DMA", "1")
    monkeypatch.setenv("TNV3_EMU_CUS", "2")
    assert _wgrad_wino_case(case, "cpu") <= 4e-6


def test_wgrad_up2x_wino_with_lds_dma_landing_late(emu, monkeypatch):
    monkeypatch.setenv("TNV3_EMU_LAZY_DMA", "1")
    _wgrad_up2x_case((1, 128, 64, 64, 4, 16), "cpu")


def test_one_launch_repack_skips_panels_only_the_eval_forward_reads(emu):
    """model.repack_wino_panels() rebuilds the stale panels that were asked for SINCE THE LAST REPACK: after a validation pass the eval-only
    panels (the F(4x4) forward panels when training runs the F(2x2) forward) are not rebuilt in front of every training step; the next eval
    forward repacks them lazily.  Block-level, no forward needed."""
    from tracknetv3_amd import ops
    from tracknetv3_amd.model import Conv2DBlock, TrackNet
    m = TrackNet(9, 3)
    blocks = [b for b in m.modules() if isinstance(b, Conv2DBlock) and b.conv.out_dim % 64 == 0 and b.conv.in_dim >= 16][:3]
    sizes, real = [], ops.pack_wino_weights_multi

    def counting(specs, variant=None):
        sizes.append(len(specs))
        return real(specs, variant)
    ops.pack_wino_weights_multi = counting
    try:
        def bump():                                  # what an optimiser step does to the version counters
            with torch.no_grad():
                for b in blocks:
                    b.conv.weight.add_(0.0)

        def train_use():
            return [(b.packed_wino(), b.packed_wino43_t()) for b in blocks]

        def eval_use():
            return [b.packed_wino43() for b in blocks]
        train_use()                                   # step 1: lazily packed, planned
        bump()
        assert m.repack_wino_panels() == 6 and sizes == [6]
        train_use()
        bump()
        eval_use()                                    # validation: eval panels packed one by one, fresh
        assert m.repack_wino_panels() == 6            # the training panels only
        before = [b._cache[("w43", 0, ops.wino43_variant(None))][1].clone() for b in blocks]
        train_use()
        bump()
        assert m.repack_wino_panels() == 6            # eval panels are stale now, but nobody asked for them since the last repack
        train_use()
        with torch.no_grad():
            blocks[0].conv.weight.mul_(2.0)
        new = eval_use()                              # the next validation repacks them lazily, from the current weights
        assert torch.equal(new[0], 2.0 * before[0]) and torch.equal(new[1], before[1])
        assert m.repack_wino_panels() == 2            # ... and the one block whose weight moved: its two training panels
        assert sizes == [6, 6, 6, 2]
    finally:
        ops.pack_wino_weights_multi = real


def _dest_views(shapes, guard=-777.0):
    """A flat buffer as the data-parallel reducer lays its buckets out (slots 64 floats apart at least, guard values between them)."""
    offs, off = [], 64
    for shp in shapes:
        offs.append(off)
        off += (int(np.prod(shp)) + 63) // 64 * 64 + 64
    flat = torch.full((off,), guard)
    return flat, [flat[o:o + int(np.prod(s))].view(s) for o, s in zip(offs, shapes)]


def _guards_intact(flat, views, guard=-777.0):
    mask = torch.ones(flat.numel(), dtype=torch.bool)
    base = flat.data_ptr()
    for v in views:
        o = (v.data_ptr() - base) // 4
        mask[o:o + v.numel()] = False
    return bool((flat[mask] == guard).all())


def test_gradient_kernels_write_into_caller_named_destinations_emulated(emu):
    """VERDICT r4 #3c: every kernel that produces a parameter gradient (the three weight-gradient entries, BatchNorm backward's
    dgamma / dbeta, both head backwards) writes it where the caller says -- the data-parallel reducer names the parameter's view of its
    flat all-reduce bucket (parallel.GradAllReducer.dest) -- bit-identically to the call that allocates, touching nothing around it."""
    from tracknetv3_amd import _lib, ops
    # plain layer, Winograd forms (F(4x4) kernel 8 and the library's F(2x2) pick) and the direct kernel
    n, cin, cout, h, w = 1, 64, 64, 4, 16
    x, dz = torch.relu(T((n, cin, h, w), 51)), T((n, cout, h, w), 52)
    for call in (lambda **kw: ops.conv3x3_wgrad_wino(x, dz, variant=8, **kw), lambda **kw: ops.conv3x3_wgrad_wino(x, dz, variant=1, **kw),
                 lambda **kw: ops.conv3x3_wgrad(x, dz, **kw)):
        want = call()
        flat, (v,) = _dest_views([(cout, cin, 3, 3)])
        got = call(out=v)
        assert got.data_ptr() == v.data_ptr() and torch.equal(got, want) and _guards_intact(flat, [v])
    # decoder entry
    xl, skip, dz2 = T((1, 128, 4, 16), 31), T((1, 64, 8, 32), 32), T((1, 64, 8, 32), 33)
    want = ops.conv3x3_wgrad_up2x(xl, skip, dz2)
    flat, (v,) = _dest_views([(64, 192, 3, 3)])
    got = ops.conv3x3_wgrad_up2x(xl, skip, dz2, out=v)
    assert got.data_ptr() == v.data_ptr() and torch.equal(got, want) and _guards_intact(flat, [v])
    # BatchNorm + ReLU backward
    c = 70
    z, da = T((3, c, 4, 8), 1, -2, 3), T((3, c, 4, 8), 6)
    g, b = T((c,), 2, 0.5, 1.5), T((c,), 3)
    a, mean, invstd = ops.bn_train_forward(z, g, b, T((c,), 4), T((c,), 5, 0.5, 2.0))
    dz_w, dg_w, db_w = ops.bn_relu_backward(da.clone(), a, z, g, mean, invstd)
    flat, (vg, vb) = _dest_views([(c,), (c,)])
    dz_g, dg, db = ops.bn_relu_backward(da.clone(), a, z, g, mean, invstd, out=(vg, vb))
    assert dg.data_ptr() == vg.data_ptr() and db.data_ptr() == vb.data_ptr() and _guards_intact(flat, [vg, vb])
    assert torch.equal(dz_g, dz_w) and torch.equal(dg, dg_w) and torch.equal(db, db_w)
    # head backward, plain and fused with WBCE
    l = 3
    ah, wt, bias = T((2, 64, 4, 16), 71), T((l, 64, 1, 1), 72, -0.3, 0.3), T((l,), 73)
    p = ops.head1x1_sigmoid(ah, wt, bias)
    dp = T((2, l, 4, 16), 74)
    y = (T((2, l, 4, 16), 75) > 0.8).float()
    up = torch.ones(1)
    for call in (lambda **kw: ops.head_backward(dp, p, ah, wt, **kw), lambda **kw: ops.head_wbce_backward(y, p, ah, wt, up, True, **kw)):
        da_w, dw_w, dbb_w = call()
        flat, (vw, vbb) = _dest_views([(l, 64, 1, 1), (l,)])
        da_g, dw_g, dbb_g = call(out=(vw, vbb))
        assert dw_g.data_ptr() == vw.data_ptr() and dbb_g.data_ptr() == vbb.data_ptr() and _guards_intact(flat, [vw, vbb])
        assert torch.equal(da_g, da_w) and torch.equal(dw_g, dw_w) and torch.equal(dbb_g, dbb_w)
    # a destination of the wrong size, type or alignment is refused, not written
    flat, (v,) = _dest_views([(cout, cin, 3, 3)])
    for bad in (v.reshape(-1)[:-1], v.double(), flat[65:65 + cout * cin * 9]):
        with pytest.raises(_lib.Tnv3Error):
            ops.conv3x3_wgrad_wino(x, dz, out=bad)


@pytest.mark.parametrize("plain_variant", [1, 2, 5, 8, -1])
def test_wgrad_up2x_follows_every_dispatchable_plain_variant_emulated(emu, monkeypatch, plain_variant):
    """tuning.WGRAD_WINO_VARIANT names the PLAIN layers' kernel; the decoder-entry weight gradient must run for each
    dispatchable value (it crashed training for 2: the C entry has no skip-half kernel 2; since ABI 6 kernel 2 is a twin of libtnv3_diag.so --
    which the emulator dispatches -- and the environment knob refuses it when tuning.py is imported)."""
    from tracknetv3_amd import ops, tuning
    monkeypatch.setattr(tuning, "WGRAD_WINO_VARIANT", plain_variant)
    xl, skip, dz = T((1, 128, 4, 16), 31), T((1, 64, 8, 32), 32), T((1, 64, 8, 32), 33)
    wd = T((64, 192, 3, 3), 34, -0.3, 0.3).double().requires_grad_(True)
    x = torch.cat([xl.repeat_interleave(2, 2).repeat_interleave(2, 3), skip], 1)
    F.conv2d(x.double(), wd, padding=1).backward(dz.double())
    dw = ops.conv3x3_wgrad_up2x(xl, skip, dz)                        # wino_variant=None: follows the tuning knob
    assert rel_err(dw, wd.grad) <= 8e-6
    dwp = ops.conv3x3_wgrad_wino(torch.relu(skip), dz)               # and the plain entry accepts the same knob
    assert dwp.shape == (64, 64, 3, 3)
