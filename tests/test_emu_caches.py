"""CPU (emulator): the packed-operand caches of Conv2DBlock follow weight / statistics updates (stale
gamma/sqrt(var+eps) after a train-mode forward; `.data` writes; broadcast into replicas that already ran a forward)."""
import torch

from oracle import prng


def T(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(shape, seed, lo, hi))


def _block(seed=3):
    from tracknetv3_amd.model import Conv2DBlock
    blk = Conv2DBlock(16, 64).eval()
    with torch.no_grad():
        blk.conv.weight.copy_(T((64, 16, 3, 3), seed, -0.2, 0.2))
        blk.bn.weight.copy_(T((64,), seed + 1, 0.5, 1.5))
        blk.bn.bias.copy_(T((64,), seed + 2))
        blk.bn.running_mean.copy_(T((64,), seed + 3))
        blk.bn.running_var.copy_(T((64,), seed + 4, 0.5, 2.0))
    return blk


def _fresh_like(blk):
    from tracknetv3_amd.model import Conv2DBlock
    f = Conv2DBlock(16, 64).eval()
    f.load_state_dict(blk.state_dict())
    return f


def test_eval_scale_follows_a_train_mode_bn_forward(emu):
    """eval -> train-mode BN forward (running stats rewritten through raw pointers, as under no_grad recalibration) -> eval
    must equal a fresh module holding the same state_dict."""
    from tracknetv3_amd import ops
    blk = _block()
    x = T((1, 16, 8, 64), 11, 0.0, 1.0)
    y0 = blk(x)
    z = ops.conv3x3(x, blk.packed_weight(), 64)
    bn = blk.bn
    ops.bn_train_forward(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum)
    bn.num_batches_tracked.add_(1)                 # what the training forward does (autograd_ops.block_fwd)
    y1 = blk(x)
    assert not torch.equal(y0, y1)
    assert torch.equal(y1, _fresh_like(blk)(x))


def test_packed_filters_follow_in_place_updates_and_data_writes_need_invalidate(emu):
    blk = _block(seed=20)
    x = T((1, 16, 4, 64), 12, 0.0, 1.0)
    blk(x)
    with torch.no_grad():
        blk.conv.weight.detach().mul_(1.5)         # detach() shares the version counter: tracked
    assert torch.equal(blk(x), _fresh_like(blk)(x))
    blk.conv.weight.data.mul_(0.5)                 # .data: no version bump -> documented: call invalidate_caches()
    blk.invalidate_caches()
    assert torch.equal(blk(x), _fresh_like(blk)(x))


def test_tracknet_invalidate_caches_clears_every_block(emu):
    from tracknetv3_amd.model import TrackNet, Conv2DBlock
    m = TrackNet(9, 3)
    blocks = [b for b in m.modules() if isinstance(b, Conv2DBlock)]
    assert len(blocks) == 17
    for b in blocks:
        b._cache["probe"] = 1
    m.invalidate_caches()
    assert all(not b._cache for b in blocks)
