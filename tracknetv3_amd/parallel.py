"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl")
gradient all-reduce over xGMI, overlapped with the rest of backward on a side HIP stream.

The reference has NO multi-GPU path; the semantics are defined here and pinned by tests:
  * every rank holds a full replica; the global minibatch is sharded, rank r draws its own mixup lambdas/permutation
    over its LOCAL shard (train.py:32-37 applied per shard);
  * BatchNorm statistics are LOCAL to a rank (per-rank batch of 10 == the reference's single-GPU batch; no SyncBN);
    running stats are not synchronised -- rank 0's are the ones checkpointed (DDP convention);
  * gradients are averaged: 8-GPU step == CPU oracle that runs the shards sequentially, averages the gradient sets
    and applies one optimiser step.
The one collective: all-reduce(sum) of 53 gradient tensors = 11 341 000 fp32 = 45.4 MB per step, packed into a few
flat buckets in gradient-ready order (head first).  xGMI is point-to-point, so a ring all-reduce is per-link bound
(2*(7/8)*45.4 MB / 153 GB/s ~ 0.5 ms) -- two orders of magnitude below the ~60 ms backward it hides under.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import autograd_ops, ops


class GradAllReducer:
    """Packs gradients into flat buckets as they become final and all-reduces each full bucket asynchronously."""

    def __init__(self, params_in_ready_order, group=None, bucket_bytes=12 << 20, record_timing=False):
        self.group = group
        self.record_timing = bool(record_timing)      # per-bucket launch / finish events on the side stream (bench.py: overlap evidence)
        self.timeline = []                            # [(bucket, bytes, start_event, end_event)] of the last step
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = list(params_in_ready_order)
        dev = self.params[0].device
        self.device = dev
        self.use_stream = dev.type == "cuda"
        self.side = torch.cuda.Stream(device=dev) if self.use_stream else None
        # greedy bucket assignment in ready order
        self.slot = {}
        self.buckets = []
        cur, cur_n = [], 0
        for p in self.params:
            n = p.numel()
            if cur and (cur_n + n) * 4 > bucket_bytes:
                self._close(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += n
        if cur:
            self._close(cur)
        self.pending = []
        self.arrived = [0] * len(self.buckets)
        self.copies = 0                               # gradients that had to be copied into their bucket (0 when the kernels write there)

    SLOT_ALIGN = 64      # floats: every parameter's slot starts 256-byte aligned, so a kernel can write its gradient straight into it

    def _close(self, plist):
        a = self.SLOT_ALIGN
        padded = sum((p.numel() + a - 1) // a * a for p in plist)
        flat = torch.zeros(padded, dtype=torch.float32, device=self.device)      # (the pad floats stay zero: they ride along in the all-reduce)
        b = len(self.buckets)
        off = 0
        for p in plist:
            self.slot[id(p)] = (b, off, p.numel(), tuple(p.shape))
            off += (p.numel() + a - 1) // a * a
        self.buckets.append(dict(flat=flat, count=len(plist)))

    def num_buckets(self):
        return len(self.buckets)

    def dest(self, param):
        """The parameter's view of its bucket: where backward's kernels write the gradient (autograd_ops' destination hook).  None -- "write a
        fresh tensor" -- when param.grad already lives in that slot: a second backward without zero_grad(set_to_none=True) (gradient
        accumulation) would otherwise OVERWRITE the accumulated gradient before autograd adds the same memory to itself (2 x new instead
        of old + new, silently); with a fresh tensor autograd accumulates into the bucket view as usual and on_grad sees the view again."""
        view = self.view(param)
        g = param.grad
        if g is not None and g.data_ptr() == view.data_ptr():
            return None
        return view

    def view(self, param):
        """The parameter's slot of its flat bucket, shaped like the parameter."""
        b, off, n, shape = self.slot[id(param)]
        return self.buckets[b]["flat"][off:off + n].view(shape)

    # grad-ready hook: the gradient is in the bucket already when the producing kernel wrote it there (dest), else it is copied in;
    # hand the bucket VIEW to autograd, launch the collective when the bucket is full
    def on_grad(self, param, grad):
        b, off, n, shape = self.slot[id(param)]
        flat = self.buckets[b]["flat"]
        view = flat[off:off + n].view(shape)
        if grad.data_ptr() != view.data_ptr():
            view.copy_(grad)
            self.copies += 1
        self.arrived[b] += 1
        if self.arrived[b] == self.buckets[b]["count"]:
            self._launch(b)
        return view

    def _launch(self, b):
        flat = self.buckets[b]["flat"]
        if self.world == 1:
            return
        if self.use_stream:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            for s in autograd_ops.backward_streams(self.device):     # a bucket mixes gradients made on the main and the
                self.side.wait_stream(s)                             # weight-gradient stream: wait for both
            with torch.cuda.stream(self.side):
                flat.div_(self.world)
                if self.record_timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.side)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.record_timing:
                    # Work.wait() on a GPU collective only makes the CURRENT stream (the side stream) wait for the
                    # communicator's stream -- no host block -- so the event after it marks the collective's end
                    work.wait()
                    e1.record(self.side)
                    self.timeline.append((b, flat.numel() * 4, e0, e1))
                else:
                    self.pending.append(work)
        else:
            flat.div_(self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def begin_step(self):
        self.timeline = []

    def on_backward_end(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.use_stream:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        self.arrived = [0] * len(self.buckets)

    def reset(self):
        """Forget a partially reduced step (backward raised): wait for what was launched, clear the arrival counters."""
        for w in self.pending:
            try:
                w.wait()
            except Exception:  # noqa: BLE001
                pass
        self.pending = []
        self.arrived = [0] * len(self.buckets)


def broadcast_module(net, src=0, group=None):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t.detach(), src=src, group=group)     # detach() shares the version counter: the in-place receive bumps it
    if hasattr(net, "invalidate_caches"):
        net.invalidate_caches()          # belt and braces: operands packed from the pre-broadcast weights must not survive


def shard_range(global_batch, rank, world):
    """Contiguous shard [lo, hi) of a global minibatch; sizes differ by at most one."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def draw_mixup(batch_size, alpha, rng):
    """Per-shard mixup draw (train.py:33-36): lambda ~ Beta(alpha, alpha) folded to >= 0.5, random partner permutation."""
    lamb = rng.beta(alpha, alpha, size=batch_size)
    lamb = np.maximum(lamb, 1 - lamb).astype(np.float32)
    perm = rng.permutation(batch_size).astype(np.int32)
    return lamb, perm


class TrackNetTrainer:
    """One optimiser step of train.py:84-96 on this rank's shard, data-parallel when torch.distributed is initialised.

    step(x, y): zero_grad -> (mixup) -> forward(train) -> WBCELoss -> backward (+ overlapped gradient all-reduce) ->
    optimizer.step().  Returns the loss as a DEVICE scalar: no per-step host sync (the reference's `.item()` at
    train.py:94 is what would serialise 8 GPUs).  With `device_rng` (default) the mixup draws come from the device-side Philox
    generator, and with `optim.FusedAdam` the update is one launch: a step then issues no H2D copy at all."""

    def __init__(self, net, optimizer, alpha=0.0, seed=13, group=None, bucket_bytes=12 << 20, record_timing=False, device_rng=True,
                 direct_grads=True):
        from .utils.metric import WBCELoss
        self.device_rng, self.seed, self.steps_done = bool(device_rng), int(seed), 0
        self.fused_loss = hasattr(net, "predictor") and hasattr(net, "down_block_1")      # TrackNet: loss fused into the head
        self.record_timing = bool(record_timing)
        self.direct_grads = bool(direct_grads)         # backward's kernels write the gradients into the all-reduce buckets (no copy)
        self.last_timing = None
        self.net, self.opt, self.alpha, self.loss_fn = net, optimizer, alpha, WBCELoss
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rng = np.random.RandomState(seed + 1000 * self.rank)
        broadcast_module(net, 0, group)
        self.reducer = GradAllReducer(autograd_ops.grad_ready_order(net), group, bucket_bytes,
                                      record_timing=record_timing) if self.world > 1 else None

    def step(self, x, y):
        self.opt.zero_grad(set_to_none=True)
        self.steps_done += 1
        if self.alpha > 0:
            if self.device_rng:
                # draws made ON THE DEVICE (Philox keyed by seed + rank, counter = step): no host RNG, no H2D copy, no sync
                lam_d, perm_d = ops.mixup_draw(x.shape[0], self.alpha, self.seed + 1000 * self.rank, self.steps_done, x.device)
            else:                                                    # the reference's host protocol (train.py:33-36)
                lamb, perm = draw_mixup(x.shape[0], self.alpha, self.rng)
                lam_d, perm_d = torch.from_numpy(lamb).to(x.device), torch.from_numpy(perm).to(x.device)
            x, y = ops.mixup(x, lam_d, perm_d), ops.mixup(y, lam_d, perm_d)
        timing = self.record_timing and x.is_cuda
        if self.reducer is not None:
            self.reducer.begin_step()
            autograd_ops.set_grad_ready_hook(self.reducer.on_grad, self.reducer.on_backward_end,
                                             self.reducer.dest if self.direct_grads else None)
        try:
            self.net.train()
            if self.fused_loss:
                loss, _ = autograd_ops.tracknet_forward_loss(self.net, x, y)      # sigmoid + WBCE fused into the head, both directions
            else:
                loss = self.loss_fn(self.net(x), y)                               # the reference's two calls (train.py:92-93)
            if timing:
                b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b0.record()
            loss.backward()
            if timing:
                b1.record()          # after backward's own join of the weight-gradient stream, before the optimiser
                self.last_timing = (b0, b1, list(self.reducer.timeline) if self.reducer is not None else [])
        except BaseException:
            if self.reducer is not None:
                self.reducer.reset()      # a bucket half-filled by a failed backward must not trigger early next step
            raise
        finally:
            autograd_ops.set_grad_ready_hook(None, None, None)
        self.opt.step()
        return loss.detach()

    def overlap_report(self):
        """After a synchronised step with record_timing: when each gradient bucket's all-reduce started / ended relative to
        the start of backward, and how much of the collective time ran before backward's compute had finished.  NOTE: the
        `backward_end` mark is recorded AFTER the main stream has joined the reducer's side stream (on_backward_end), so it
        is `max(compute end, last all-reduce end)`; the compute end is estimated by the last bucket's launch time."""
        if not self.last_timing:
            return None
        b0, b1, tl = self.last_timing
        rows = [{"bucket": b, "bytes": nbytes, "launch_ms": round(b0.elapsed_time(e0), 3), "finish_ms": round(b0.elapsed_time(e1), 3)}
                for b, nbytes, e0, e1 in tl]
        total = sum(r["finish_ms"] - r["launch_ms"] for r in rows)
        compute_end = max((r["launch_ms"] for r in rows), default=0.0)      # the last bucket is launched when the last gradient is final
        hidden = sum(max(0.0, min(r["finish_ms"], compute_end) - r["launch_ms"]) for r in rows)
        return {"backward_ms": round(b0.elapsed_time(b1), 3), "last_gradient_ready_ms": round(compute_end, 3), "buckets": rows,
                "allreduce_ms_total": round(total, 3), "overlap_fraction": round(hidden / total, 4) if total > 0 else None}
