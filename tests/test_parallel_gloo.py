"""CPU, world_size 2 over gloo: the data-parallel machinery -- gradient bucketing in ready order, averaging,
hook -> view replacement, shard ranges, per-rank mixup draws.  (The kernels themselves are rank-local and are
covered by the emulator / GPU suites; RCCL replaces gloo on the GPUs with the same torch.distributed calls.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tracknetv3_amd import autograd_ops, parallel
        from tracknetv3_amd.utils.general import get_model
        torch.manual_seed(100 + rank)                      # replicas start DIFFERENT on purpose
        net = get_model("TrackNet", 3, "")
        parallel.broadcast_module(net, 0)
        w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        order = autograd_ops.grad_ready_order(net)
        assert len(order) == 53 and len({id(p) for p in order}) == 53
        assert order[0] is net.predictor.weight and order[-1] is net.down_block_1.conv_1.conv.weight
        red = parallel.GradAllReducer(order, bucket_bytes=12 << 20)
        assert 3 <= red.num_buckets() <= 8
        gen = torch.Generator().manual_seed(7 + rank)
        grads, views = {}, {}
        for p in order:                                     # what backward does, in the same order
            g = torch.randn(p.shape, generator=gen)
            grads[id(p)] = g.clone()
            views[id(p)] = red.on_grad(p, g)
        red.on_backward_end()
        # every rank must now hold the mean over ranks
        for p in order:
            mine = grads[id(p)]
            other = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(other, mine)
            want = sum(other) / world
            assert torch.allclose(views[id(p)], want, atol=1e-6), "bucket average mismatch"
        # second step reuses the buckets
        for p in order:
            red.on_grad(p, torch.ones(p.shape) * (rank + 1))
        red.on_backward_end()
        assert torch.allclose(views[id(order[5])], torch.full(order[5].shape, (1 + world) / 2.0))
        lo, hi = parallel.shard_range(80, rank, world)
        lam, perm = parallel.draw_mixup(hi - lo, 0.5, np.random.RandomState(13 + 1000 * rank))
        out[rank] = dict(w0=w0[:1000].clone(), shard=(lo, hi), lam=lam, perm=perm)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_sharding():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
        assert torch.equal(r0["w0"], r1["w0"]), "broadcast_module must make replicas identical"
        assert r0["shard"] == (0, 40) and r1["shard"] == (40, 80)
        assert (r0["lam"] >= 0.5).all() and sorted(r0["perm"].tolist()) == list(range(40))
        assert not np.array_equal(r0["lam"], r1["lam"]), "each rank draws its own mixup lambdas"


def _worker8(rank, world, port, out):
    """BASELINE configs[2]'s rank count: 8 ranks, global batch 80.  Gradients are WRITTEN INTO the reducer's bucket views (what the
    backward kernels do through autograd_ops' destination hook), so no copy may happen."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tracknetv3_amd import autograd_ops, parallel
        from tracknetv3_amd.utils.general import get_model
        torch.manual_seed(100 + rank)
        net = get_model("TrackNet", 8, "concat")            # the benchmark's model: 27 -> 8
        parallel.broadcast_module(net, 0)
        order = autograd_ops.grad_ready_order(net)
        red = parallel.GradAllReducer(order, bucket_bytes=12 << 20)
        assert red.world == 8 and 3 <= red.num_buckets() <= 8
        for p in order:                                     # every slot 256 bytes into its bucket (the device allocator aligns the bucket
            v = red.dest(p)                                 # itself to 512 bytes; the host allocator to 64), no overlap
            base = red.buckets[red.slot[id(p)][0]]["flat"].data_ptr()
            assert v.shape == p.shape and v.is_contiguous() and (v.data_ptr() - base) % 256 == 0 and v.data_ptr() % 16 == 0
        spans = sorted((red.dest(p).data_ptr(), red.dest(p).data_ptr() + 4 * p.numel()) for p in order)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        gen = torch.Generator().manual_seed(7 + rank)
        checks = {}
        for step in range(2):                               # the second step reuses the buckets (no stale arrival counts)
            for p in order:
                g = torch.randn(p.shape, generator=gen)
                if p.numel() <= 4096:                       # (all-gathering the big ones 8 ways would dominate the test)
                    checks[id(p)] = g.clone()
                view = red.dest(p)
                view.copy_(g)                               # the "kernel" writes the gradient where the collective reads it
                got = red.on_grad(p, view)
                assert got.data_ptr() == view.data_ptr()
            red.on_backward_end()
            assert red.copies == 0, "a gradient written into its destination must not be copied again"
            for p in order:
                if id(p) in checks:
                    other = [torch.empty_like(checks[id(p)]) for _ in range(world)]
                    dist.all_gather(other, checks[id(p)])
                    assert torch.allclose(red.view(p), sum(other) / world, atol=1e-6), "bucket average mismatch"
        # a gradient that arrives somewhere else (a caller's own tensor) is still copied in -- and counted
        red.on_grad(order[0], torch.ones(order[0].shape))
        assert red.copies == 1
        red.reset()
        lo, hi = parallel.shard_range(80, rank, world)
        w0 = torch.cat([p.detach().reshape(-1)[:8] for p in net.parameters()])
        pads = sum(b["flat"].numel() for b in red.buckets) - sum(p.numel() for p in order)
        out[rank] = dict(shard=(lo, hi), w0=w0, buckets=red.num_buckets(), pad_floats=pads)
    finally:
        dist.destroy_process_group()


def test_eight_rank_bucket_views_and_global_batch_80():
    """The rank count and global batch of BASELINE configs[2] on the gloo stand-in (RCCL replaces it on the GPUs with the same calls)."""
    world, port = 8, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
        res = [dict(out[r]) for r in range(world)]
    assert [r["shard"] for r in res] == [(10 * r, 10 * r + 10) for r in range(world)]      # 80 = 8 x the reference's batch of 10
    for r in res[1:]:
        assert torch.equal(r["w0"], res[0]["w0"]), "broadcast_module must make the eight replicas identical"
        assert r["buckets"] == res[0]["buckets"]
    assert 0 <= res[0]["pad_floats"] < 53 * 64


def test_shard_range_covers_batch():
    from tracknetv3_amd.parallel import shard_range
    for gb in (1, 7, 10, 80, 81):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _watchdog_worker(rank, world, port, out, hang_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench._CTL["group"] = dist.new_group(backend="gloo")
    bench.RCCL_WATCHDOG_S = 2.0
    if rank == hang_rank:                                  # this rank's "collective" does not come up in time
        real = dist.all_reduce

        def slow(t, *a, **kw):
            if not kw.get("group") and not a[1:]:
                time.sleep(5.0)
            return real(t, *a, **kw)
        dist.all_reduce = slow
    rep = bench.rccl_first_contact(torch.device("cpu"), rank, world, "gloo")
    out[rank] = rep
    time.sleep(4.0)                                        # let the late all-reduce of the hung rank finish before the group goes away
    os._exit(0)


@pytest.mark.parametrize("hang_rank", [-1, 1])
def test_bench_first_contact_watchdog_reports_instead_of_hanging(hang_rank):
    """bench.py's RCCL first-contact insurance on the gloo stand-in: a healthy group reports ok with a timing; a rank whose first
    all-reduce does not answer within the watchdog makes EVERY rank report the failure (with the rank that hung) instead of hanging
    the record."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_watchdog_worker, args=(world, port, out, hang_rank), nprocs=world, join=True)
        r0, r1 = dict(out[0]), dict(out[1])
    if hang_rank < 0:
        assert r0["ok"] and r1["ok"] and r0["first_allreduce_ms"] > 0 and not r0["any_hung"]
    else:
        assert not r0["ok"] and not r1["ok"]
        assert "within 2 s" in r0["error"] and r1["hung_here"] and r0["watchdog_s"] == 2.0
        assert r0["any_hung"] and r1["any_hung"]           # every rank learns that SOMEONE hung: all of them leave without a device synchronise


def test_destination_hook_steps_aside_when_a_gradient_is_already_accumulated_in_the_bucket():
    """with the kernels writing into the bucket views, param.grad ALIASES the bucket after the first backward.  A second backward
    without zero_grad(set_to_none=True) (gradient accumulation) must not be handed the same memory as its destination -- the kernel would
    overwrite the accumulated gradient before autograd adds that memory to itself (2 x new instead of old + new).  dest() then answers None
    ("write a fresh tensor"), and autograd's own accumulation into the view stays correct."""
    from tracknetv3_amd import parallel
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    red = parallel.GradAllReducer(params, bucket_bytes=1 << 20)
    for p in params:
        v = red.dest(p)
        assert v is not None and v.data_ptr() == red.view(p).data_ptr()
        v.fill_(2.0)
        p.grad = red.on_grad(p, v)                              # what autograd does with the hook's return value
        assert red.dest(p) is None                              # a second backward now: fresh tensor + copy-free accumulation by autograd
        fresh = torch.full(p.shape, 3.0)
        p.grad.add_(fresh)                                      # AccumulateGrad
        assert torch.equal(red.view(p), torch.full(p.shape, 5.0))
        p.grad = None                                           # zero_grad(set_to_none=True)
        assert red.dest(p) is not None
