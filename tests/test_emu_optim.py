"""CPU (emulator): fused multi-tensor Adam / SGD / gradient-norm clipping against torch.optim on the same tensors, and the
device-side mixup draws (train.py:33-36, 85, 96, 165, 242-248)."""
import numpy as np
import pytest
import torch

from oracle import prng


def T(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(shape, seed, lo, hi))


SHAPES = [(64, 27, 3, 3), (64,), (64,), (5000,), (3, 4097), (1,), (128, 64, 3, 3), (8, 64, 1, 1), (8,)]


def _params(seed):
    return [torch.nn.Parameter(T(s, seed + k, -0.5, 0.5)) for k, s in enumerate(SHAPES)]


def _ulps(a, b, floor=1e-30):
    """max |a - b| in units of the fp32 spacing at max(|a|, |b|, floor).  `floor`: the magnitude of one update step -- a
    parameter that happens to sit closer to zero than its own update cannot be compared in ulps of itself."""
    a, b = a.detach().double(), b.detach().double()
    scale = torch.maximum(a.abs(), b.abs()).clamp_min(floor)
    spacing = 2.0 ** (torch.floor(torch.log2(scale)) - 23)
    return ((a - b).abs() / spacing).max().item()


@pytest.mark.parametrize("clip", [None, 0.7])
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_tracks_torch_adam_over_ten_steps(emu, clip, wd):
    from tracknetv3_amd.optim import FusedAdam
    mine, ref = _params(100), _params(100)
    o_m = FusedAdam(mine, lr=1e-3, weight_decay=wd, max_grad_norm=clip)
    o_r = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd, foreach=False)
    for it in range(10):
        for k, (a, b) in enumerate(zip(mine, ref)):
            g = T(a.shape, 1000 * it + k, -2.0, 2.0) * (10.0 ** ((k % 4) - 2))      # gradient scales 1e-2 .. 1e1
            a.grad, b.grad = g.clone(), g.clone()
        if clip is not None:
            total = torch.nn.utils.clip_grad_norm_(ref, clip, foreach=False)
        o_r.step()
        o_m.step()
        if clip is not None:
            assert abs(o_m.last_grad_norm[0].item() - total.item()) <= 1e-6 * total.item()
            for a, b in zip(mine, ref):                       # the clipped gradients are what clip_grad_norm_ leaves behind
                assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)     # the coefficient comes from an fp64 norm here, fp32 in torch
    for a, b in zip(mine, ref):
        # ulps at the scale of the parameter or of one update (lr); torch's CPU kernels round differently from its GPU ones
        # (addcdiv is (alpha * t1) / t2 on the CPU, alpha * (t1 / t2) on the GPU), so a few ulps here, <= 1 in the -m gpu twin
        assert _ulps(a, b, floor=1e-3) <= 64, (tuple(a.shape), _ulps(a, b, floor=1e-3))
        sa, sb = o_m.state[a], o_r.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 10.0
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=2e-6 * float(sb["exp_avg"].abs().max()))
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=4e-6, atol=1e-20)
    # state_dict layout is torch.optim.Adam's: checkpoints interchange (train.py:283-301)
    sd_m, sd_r = o_m.state_dict(), o_r.state_dict()
    assert sd_m["state"].keys() == sd_r["state"].keys()
    assert all(set(sd_m["state"][k]) == set(sd_r["state"][k]) for k in sd_m["state"])
    o_r2 = torch.optim.Adam(_params(100), lr=1e-3, weight_decay=wd)
    o_r2.load_state_dict(sd_m)


def test_fused_adam_bumps_versions_and_zeroes_grads(emu):
    from tracknetv3_amd.optim import FusedAdam
    ps = _params(5)[:3]
    opt = FusedAdam(ps, zero_grad_in_step=True)
    v0 = [p._version for p in ps]
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    assert all(p._version > v for p, v in zip(ps, v0))
    assert all(float(p.grad.abs().sum()) == 0.0 for p in ps)


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_fused_sgd_tracks_torch_sgd(emu, momentum):
    from tracknetv3_amd.optim import FusedSGD
    mine, ref = _params(7), _params(7)
    o_m = FusedSGD(mine, lr=1e-3, momentum=momentum, weight_decay=1e-4)
    o_r = torch.optim.SGD(ref, lr=1e-3, momentum=momentum, weight_decay=1e-4, foreach=False)
    for it in range(6):
        for k, (a, b) in enumerate(zip(mine, ref)):
            g = T(a.shape, 77 * it + k, -1.0, 1.0)
            a.grad, b.grad = g.clone(), g.clone()
        o_r.step()
        o_m.step()
    for a, b in zip(mine, ref):
        assert _ulps(a, b, floor=1e-3) <= 4
    if momentum:
        assert all(torch.allclose(o_m.state[a]["momentum_buffer"], o_r.state[b]["momentum_buffer"], rtol=1e-6, atol=1e-12) for a, b in zip(mine, ref))


def test_more_than_64_tensors_and_argument_errors(emu):
    from tracknetv3_amd import _lib, ops
    ps = [T((37 + k,), k) for k in range(70)]
    gs = [T((37 + k,), 100 + k) for k in range(70)]
    m, v = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    for r, g in zip(ref, gs):
        r.grad = g.clone()
    torch.optim.Adam(ref, lr=1e-2, foreach=False).step()
    ops.adam_step(ps, gs, m, v, 1, lr=1e-2)
    assert max(_ulps(a, b, floor=1e-2) for a, b in zip(ps, ref)) <= 4
    nc = ops.grad_norm(gs, 1.0)
    want = torch.sqrt(sum((g.double() ** 2).sum() for g in gs))
    assert abs(nc[0].item() - want.item()) <= 1e-6 * want.item() and abs(nc[1].item() - 1.0 / (want.item() + 1e-6)) <= 1e-6
    with pytest.raises(_lib.Tnv3Error):
        ops.adam_step(ps, gs, m, v, 0)                        # the step counts from 1
    with pytest.raises(_lib.Tnv3Error):
        ops.adam_step(ps, gs[:-1], m, v, 1)


def test_mixup_draws_on_the_device(emu):
    """Determinism per (seed, step), a valid permutation, lambda in [0.5, 1) with the moments of the folded Beta(alpha, alpha)."""
    from tracknetv3_amd import ops
    lam, perm = ops.mixup_draw(10, 0.5, 13, 1, "cpu")
    lam2, perm2 = ops.mixup_draw(10, 0.5, 13, 1, "cpu")
    assert torch.equal(lam, lam2) and torch.equal(perm, perm2)
    lam3, perm3 = ops.mixup_draw(10, 0.5, 13, 2, "cpu")
    assert not torch.equal(lam, lam3)
    assert sorted(perm.tolist()) == list(range(10)) and sorted(perm3.tolist()) == list(range(10))
    for alpha in (0.5, 1.0, 2.0):
        big, p = ops.mixup_draw(20000, alpha, 7, 5, "cpu")
        assert sorted(p.tolist()) == list(range(20000))
        x = big.double().numpy()
        assert x.min() >= 0.5 and x.max() <= 1.0
        rng = np.random.RandomState(3)
        ref = rng.beta(alpha, alpha, size=400000)
        ref = np.maximum(ref, 1 - ref)
        assert abs(x.mean() - ref.mean()) <= 4e-3 and abs(x.std() - ref.std()) <= 4e-3, (alpha, x.mean(), ref.mean(), x.std(), ref.std())
        q = np.quantile(x, [0.1, 0.5, 0.9]) - np.quantile(ref, [0.1, 0.5, 0.9])
        assert np.abs(q).max() <= 8e-3, (alpha, q)
    # positions of element 0 over many permutations of 10 are uniform
    pos = np.zeros(10)
    for s in range(400):
        _, pm = ops.mixup_draw(10, 0.5, 99, s, "cpu")
        pos[pm.tolist().index(0)] += 1
    assert pos.min() >= 15 and pos.max() <= 70, pos


def test_fused_sgd_writes_the_clipped_gradient_back(emu):
    """clip_grad_norm_ (train.py:165) modifies the gradients in place: after a clipped FusedSGD step `.grad` holds grad * coef."""
    from tracknetv3_amd.optim import FusedSGD
    mine, ref = _params(300), _params(300)
    o_m = FusedSGD(mine, lr=1e-2, momentum=0.9, max_grad_norm=0.5)
    o_r = torch.optim.SGD(ref, lr=1e-2, momentum=0.9, foreach=False)
    for it in range(3):
        for k, (a, b) in enumerate(zip(mine, ref)):
            g = T(a.shape, 7000 + 50 * it + k, -2.0, 2.0)
            a.grad, b.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.5, foreach=False)
        o_r.step()
        o_m.step()
        for a, b in zip(mine, ref):
            assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)
            assert _ulps(a, b, floor=1e-2) <= 16


def test_fused_adam_validates_up_front_and_keeps_counters_on_failure(emu):
    from tracknetv3_amd.optim import FusedAdam
    ps = _params(400)
    with pytest.raises(ValueError, match="beta1"):
        FusedAdam(ps, betas=(0.4, 0.999))                      # refused in __init__, not at the first step
    opt = FusedAdam(ps, lr=1e-3)
    for k, p in enumerate(ps):
        p.grad = T(p.shape, 8000 + k)
    opt.step()
    assert all(float(opt.state[p]["step"]) == 1.0 for p in ps)
    opt.param_groups[0]["decoupled_weight_decay"] = True       # a later edit is caught at step time ...
    before = [p.detach().clone() for p in ps]
    with pytest.raises(RuntimeError, match="decoupled_weight_decay"):
        opt.step()
    assert all(float(opt.state[p]["step"]) == 1.0 for p in ps)          # ... before any counter or tensor moved
    assert all(torch.equal(a, b) for a, b in zip(before, ps))
    opt.param_groups[0]["decoupled_weight_decay"] = False
    opt.param_groups[0]["betas"] = (0.3, 0.999)
    with pytest.raises(ValueError, match="beta1"):
        opt.step()
    assert all(float(opt.state[p]["step"]) == 1.0 for p in ps)


def test_device_guard_sees_tensors_inside_list_arguments():
    """ops.grad_norm / adam_step / sgd_step / inpaintnet_pack take LISTS of tensors: the device guard must find them."""
    from tracknetv3_amd import _lib, ops
    a, b = torch.zeros(3), torch.ones(2)
    assert _lib.first_tensor((a, [b])) is a
    assert _lib.first_tensor(([b, a], 1.0)) is b
    assert _lib.first_tensor(((), [], (b,))) is b
    assert _lib.first_tensor((1, "x", None)) is None
    for name in ("grad_norm", "adam_step", "sgd_step", "inpaintnet_pack"):
        assert hasattr(getattr(ops, name), "__wrapped__"), name          # wrapped by _lib.on_tensor_device
