#!/usr/bin/env python
"""bench.py -- TrackNetV3 hot-path benchmark on MI355X (contract: see the build brief / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N ...`: WORLD_SIZE / RANK / LOCAL_RANK in the environment) or -- plain `python bench.py --gpus N` -- this
script re-executes itself under torch.distributed.run with N ranks.  Either way it REFUSES to run with fewer ranks or GPUs
than `--gpus` says (exit code != 0), and the line carries `config.rccl_world_size` read back from the process group.

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
BASELINE.json configs[1] -- TrackNet(seq_len=8, bg_mode='concat').eval() forward, batch 10 per GPU, 288x512 --
i.e. 80 output heat-map frames per GPU-step.  Weak scaling: every rank runs its own batch; inference windows are
independent, so there is no data-path collective.  Rank 0 prints ONE JSON line.

Extra modes:  --tune  (time every compiled conv tile configuration per layer shape, write conv_tuning.json)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SEQ_LEN, BG_MODE, H, W = 8, "concat", 288, 512
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix peak (= fp32 vector peak)
PEAK_HBM_GBPS = 8000.0
KERNEL_SET = "wino43s+up2x_wino43s"   # the conv kernel families of the eval forward (profiles/conv_traffic.json must match, and so must the library)


def lib_sha256():
    """sha256 of the libtnv3_hip.so this process loaded: counter files (profiles/conv_traffic.json) are replayed only for the same build."""
    import hashlib
    from tracknetv3_amd import _build
    try:
        with open(_build.LIB, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def src_sha256():
    """Path-independent key of the build: sha256 of the kernel sources + compile flags (tracknetv3_amd/_build.source_sha256)."""
    from tracknetv3_amd import _build
    try:
        return _build.source_sha256()
    except OSError:
        return None


ALG_BYTES_PER_SAMPLE = 693.55e6    # ideal-fusion fp32 bytes of one 27->8 forward


def conv_layer_table(in_dim, h, w):
    """[(name, c0, c1, cout, H, W, up)] -- the 17 Conv2DBlocks of TrackNet.forward (model.py:57-73)."""
    t = [("down_block_1.conv_1", in_dim, 0, 64, h, w, False), ("down_block_1.conv_2", 64, 0, 64, h, w, False),
         ("down_block_2.conv_1", 64, 0, 128, h // 2, w // 2, False), ("down_block_2.conv_2", 128, 0, 128, h // 2, w // 2, False),
         ("down_block_3.conv_1", 128, 0, 256, h // 4, w // 4, False), ("down_block_3.conv_2", 256, 0, 256, h // 4, w // 4, False),
         ("down_block_3.conv_3", 256, 0, 256, h // 4, w // 4, False),
         ("bottleneck.conv_1", 256, 0, 512, h // 8, w // 8, False), ("bottleneck.conv_2", 512, 0, 512, h // 8, w // 8, False),
         ("bottleneck.conv_3", 512, 0, 512, h // 8, w // 8, False),
         ("up_block_1.conv_1", 512, 256, 256, h // 4, w // 4, True), ("up_block_1.conv_2", 256, 0, 256, h // 4, w // 4, False),
         ("up_block_1.conv_3", 256, 0, 256, h // 4, w // 4, False),
         ("up_block_2.conv_1", 256, 128, 128, h // 2, w // 2, True), ("up_block_2.conv_2", 128, 0, 128, h // 2, w // 2, False),
         ("up_block_3.conv_1", 128, 64, 64, h, w, True), ("up_block_3.conv_2", 64, 0, 64, h, w, False)]
    return t


def conv_flops(c0, c1, cout, h, w):
    return 2.0 * 9 * (c0 + c1) * cout * h * w


def tune(dev, batch, out_paths):
    """Time every compiled tile configuration on every distinct conv shape of the workload; keep the fastest."""
    from tracknetv3_amd import ops, tuning
    in_dim = (SEQ_LEN + 1) * 3
    ncfg = ops.conv3x3_num_configs()
    infos = [ops.conv3x3_config_info(c) for c in range(ncfg)]
    best, report, seen = {}, [], set()
    shapes = list(conv_layer_table(in_dim, H, W))
    # data-gradient convolutions of the training step: dX = conv(dZ, W^T): Cout' = Cin, Cin' = Cout, single source
    shapes += [(name + ":dgrad", cout, 0, c0 + c1, h, w, False) for (name, c0, c1, cout, h, w, up) in shapes[1:]]
    for name, c0, c1, cout, h, w, up in shapes:
        key = f"{cout},{c0 + c1},{batch},{h},{w}"
        if key in seen:
            continue
        seen.add(key)
        wt = torch.empty(cout, c0 + c1, 3, 3, device=dev).uniform_(-0.05, 0.05)
        wp = ops.pack_conv3x3_weights(wt)
        s0 = torch.rand((batch, c0, h // 2, w // 2) if up else (batch, c0, h, w), device=dev)
        s1 = torch.rand((batch, c1, h, w), device=dev) if c1 else None
        sc, sh, mu = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev) - 0.5, torch.rand(cout, device=dev) - 0.5
        out = torch.empty((batch, cout, h, w), device=dev)
        fl = conv_flops(c0, c1, cout, h, w) * batch
        row = {"layer": name, "key": key, "tflops": {}}
        for cfg in range(ncfg):
            if cout % infos[cfg]["m_block"] or (c1 and c0 % infos[cfg]["chan_chunk"]):
                continue
            for _ in range(2):
                ops.conv3x3(s0, wp, cout, src1=s1, mean=mu, scale=sc, shift=sh, up0=up, relu=True, cfg=cfg, out=out)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                ops.conv3x3(s0, wp, cout, src1=s1, mean=mu, scale=sc, shift=sh, up0=up, relu=True, cfg=cfg, out=out)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            row["tflops"][str(cfg)] = round(fl / ms / 1e9, 2)
        cfg_best = max(row["tflops"], key=lambda k: row["tflops"][k])
        best[key] = int(cfg_best)
        row["best"] = int(cfg_best)
        report.append(row)
        print(f"[tune] {name:22s} {key:24s} best cfg {cfg_best}: {row['tflops']}", file=sys.stderr, flush=True)
        del wt, wp, s0, s1, out
    meta = {"device": torch.cuda.get_device_name(dev), "batch": batch, "configs": infos}
    tuning.save(best, meta)
    for p in out_paths:
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            json.dump({"meta": meta, "configs": best, "report": report}, f, indent=1)
    return best


def _cpu_info():
    """CPU model, physical cores (unique (physical id, core id) pairs) and logical CPUs from /proc/cpuinfo."""
    model, cores, sockets = None, set(), set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model is None:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                    sockets.add(phys)
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return {"cpu_model": model, "physical_cores": len(cores) or None, "sockets": len(sockets) or None, "logical_cpus": os.cpu_count()}


def cpu_baseline(mode="infer", budget_s=40.0):
    """the oracle's PyTorch-CPU restatement of the same path on the GPU box's host cores, fp32, the
    same kind of synthetic tensors, 2 warm-up iterations, median of >= 5 timed ones, CPU model / physical cores stated, and a
    1-thread figure beside the best multi-thread one.
    * infer: configs[1] as written -- eval forward, batch 10, 288x512 (80 frames per run).
    * train: configs[2] shard reduced to batch 2 (a bounded sample: a batch-10 CPU step takes ~1 min) -- mixup with injected
      draws, forward in train mode, WBCE, backward (autograd), torch.optim.Adam(lr=1e-3) step on the 53 tensors (train.py:84-96).
    Threads: a short scan at the measured batch over {16, 32, 64, physical cores} keeps the fastest -- running on every SMT thread of a
    2-socket host is several times SLOWER for these shapes (oneDNN), which would flatter the GPU.  Returns frames/s."""
    from oracle import nets
    in_dim = (SEQ_LEN + 1) * 3
    info = _cpu_info()
    ncpu = os.cpu_count() or 1
    phys = info["physical_cores"] or max(1, ncpu // 2)
    sd = nets.synth_state(nets.tracknet_state_shapes(in_dim, SEQ_LEN), 31, calibrated=(mode == "infer"))
    t_all = time.time()

    def fwd(x):
        t = time.time()
        with torch.no_grad():
            nets.tracknet_forward(sd, x, training=False)
        return time.time() - t

    opt_state = {}

    def train_step(x, y):
        t = time.time()
        lam = np.linspace(0.55, 0.95, x.shape[0])
        xm, ym = nets.mixup_injected(x, y, lam, list(range(x.shape[0]))[::-1])
        _, _, grads, _ = nets.tracknet_train_step_grads(sd, xm, ym, torch.float32)
        if "opt" not in opt_state:                      # the optimiser step of train.py:96 on the oracle's own tensors
            opt_state["names"] = list(grads.keys())
            opt_state["opt"] = torch.optim.Adam([sd[k] for k in opt_state["names"]], lr=1e-3)
        for k in opt_state["names"]:
            sd[k].grad = grads[k].to(sd[k].dtype)
        opt_state["opt"].step()
        return time.time() - t

    n = 10 if mode == "infer" else 2
    xn = nets.synth_input((n, in_dim, H, W), 4244)
    yn = nets.disc_heatmaps(n, SEQ_LEN, H, W, 4243) if mode == "train" else None
    run = (lambda: fwd(xn)) if mode == "infer" else (lambda: train_step(xn, yn))
    scan = {}
    torch.set_num_threads(min(16, ncpu))
    run()                                       # warm-up (thread pool, oneDNN primitive cache)
    cand = sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu), min(phys, ncpu)})
    for th in cand:                             # scan at the MEASURED batch size; the physical-core count is always tried
        over = scan and (time.time() - t_all > budget_s * 0.5 or scan[max(scan)] > 1.5 * min(scan.values()))
        if over and th != min(phys, ncpu):
            continue                            # out of budget, or clearly past the sweet spot (more threads = slower)
        if over and time.time() - t_all > budget_s * 0.8:
            continue
        torch.set_num_threads(th)
        scan[th] = run()
    best = min(scan, key=scan.get)
    torch.set_num_threads(best)
    run(); run()                                # 2 warm-up iterations
    times = [run() for _ in range(5)]
    while time.time() - t_all < budget_s and len(times) < 9:
        times.append(run())
    med = float(np.median(times))
    one = None
    if mode == "infer":                         # 1-thread figure (scaling context): batch 1, one warm-up + two timed runs
        torch.set_num_threads(1)
        x1 = nets.synth_input((1, in_dim, H, W), 4245)
        fwd(x1)
        t1 = float(np.median([fwd(x1), fwd(x1)]))
        torch.set_num_threads(best)
        one = {"value": round(SEQ_LEN / t1, 3), "unit": "frames/s", "sample": f"batch 1, median of 2 runs ({t1:.1f} s each)"}
    what = "eval forward" if mode == "infer" else ("train step (mixup + forward(train) + WBCE + backward + Adam step; a BOUNDED sample: "
                                                   "batch 2 instead of 10 -- a batch-10 CPU step takes about a minute and the thread "
                                                   "scan alone would exhaust the leg's budget; frames/s is per-sample work, so the "
                                                   "figure is comparable)")
    out = {"value": round(n * SEQ_LEN / med, 2), "unit": "frames/s", "cores": best, "kind": "port",
           "sample": f"oracle (torch-CPU fp32 restatement, equal to the imported reference) TrackNet(27,8) {what}, batch {n} x 288x512, "
                     f"2 warm-ups, median of {len(times)} runs ({med:.2f} s each) on {best} threads; thread scan at that batch, s/run: "
                     f"{{{', '.join(f'{k}: {v:.2f}' for k, v in scan.items())}}}",
           "one_thread": one, "torch": torch.__version__}
    out.update(info)
    return out


TRAIN_FLOPS_PER_SAMPLE = 678.2e9    # fwd + dgrad + wgrad (no dgrad for the first layer)
# What the matrix pipe executes per sample (GFLOP): the upsampled halves of up_block_{1,2,3}.conv_1 cost 4/9 in all three
# passes (conv_up2x / dgrad_up2x / 2x2-window wgrad); the plain halves with >= 24 input channels run in Winograd F(2x2,3x3)
# form (16/36) in forward and data gradient, and those with >= 64 channels on both sides also in the weight gradient
# (everything but the first layer).
#   forward 98.7 (of 227.6), data gradient 96.6 (of 223.0), weight gradient 101.2 (of 227.6)
# Round 2: all three passes of the upsampled halves run in Winograd forms that keep 9 of the 16 GEMMs (9/36 instead of 4/9 of
# their 65.2 GFLOP each): forward 86.0, data gradient 83.9, weight gradient 88.5.
# Round 3: the stem's weight gradient also runs in Winograd form, on a 64-channel block of which 27 channels are live: the pipe executes
# 4.83 GFLOP there instead of the direct kernel's 4.59 -- 0.1 % of the total, the constant stays.
TRAIN_FLOPS_EXECUTED_PER_SAMPLE_CLASS_FILTERS = 296.5e9
TRAIN_FLOPS_EXECUTED_PER_SAMPLE = 296.5e9 - 3 * 65.2e9 * (4 / 9 - 9 / 36)


def train_flops_executed_per_sample():
    """What the matrix pipe executes per sample in one training step with the kernels the step dispatches to (tuning.*): per layer and
    pass the direct count times 9/36 (Winograd F(4x4,3x3), and the 9-GEMM forms of the upsampled halves), 16/36 (F(2x2,3x3)) or 1."""
    from tracknetv3_amd import tuning as t
    total = 0.0
    for idx, (name, c0, c1, co, h, w, up) in enumerate(conv_layer_table((SEQ_LEN + 1) * 3, H, W)):      # idx: the layer's forward order (tuning.WINO43_TRAIN_F22_LAYERS)
        def frac(ci, use43):
            if use43:
                return 9 / 36
            return 16 / 36 if t.use_winograd(ci, co, h, w) else 1.0

        def wfrac(ci):                                   # weight gradient: F(4x4) kernel 8, an F(2x2) kernel, or the direct form
            if t.use_wino43_wgrad(ci, co, h, w):
                return 9 / 36
            return 16 / 36 if t.use_winograd_wgrad(ci, co, h, w) else 1.0
        if up:
            from tracknetv3_amd import ops as _ops
            uvt = _ops.up2x_wino_variant(t.UP2X_WINO_VARIANT_TRAIN)
            f_fwd = 6.25 / 36 if (uvt == 2 and _ops.up2x_wino_supported(c0, co, h // 2, w // 2, 2)) else 9 / 36
            hl, wl = h // 2, w // 2
            dv = _ops.dgrad_up2x_wino_variant()
            f_dg = 6.25 / 36 if (dv == 2 and _ops.dgrad_up2x_wino_supported(c0, co, hl, wl, 2)) else 9 / 36
            wv = int(t.WGRAD_UP2X_VARIANT)
            f_wg = 6.25 / 36 if ((wv < 0 or wv == 2) and co % 64 == 0 and hl % 2 == 0 and wl % 8 == 0) else (9 / 36 if (wv != 0 and c0 % 128 == 0) else 16 / 36)
            upf = conv_flops(c0, 0, co, h, w) * (f_fwd + f_dg + f_wg)      # forward; data gradient; weight gradient (25-of-36 F(4x4) or 9-GEMM F(2x2) forms)
            sk = conv_flops(c1, 0, co, h, w)
            skip = sk * (frac(c1, t.use_wino43_train(c1, co, h, w, layer=idx)) + frac(c1, t.use_wino43_dgrad(co, c1, h, w)) + wfrac(c1))
            total += upf + skip
        else:
            fl = conv_flops(c0, 0, co, h, w)
            total += fl * frac(c0, t.use_wino43_train(c0, co, h, w, layer=idx))                     # forward
            if name != "down_block_1.conv_1":
                total += fl * frac(c0, t.use_wino43_dgrad(co, c0, h, w))                            # data gradient (none for the first layer)
            total += fl * wfrac(c0)                                                                 # weight gradient
    return total


STRONG_GLOBAL_BATCH = 80            # BASELINE configs[2]: global batch 80 = 8 GPUs x the reference's --batch_size 10 (README.md:144)


def train_leg(dev, rank, world, batch, steps, warmup, record_timing=False, strong_steps=3):
    """BASELINE configs[2] shard: TrackNet(27,8) train step, batch 10 per GPU, mixup alpha 0.5, WBCE, backward, Adam, DP gradient
    all-reduce over RCCL when world > 1.  Times `steps` steps between barriers, max over ranks; returns the JSON fields (on
    every rank; rank 0 prints).  record_timing: one extra, synchronised step with per-bucket all-reduce events (overlap report).
    strong_steps > 0: the same step at a FIXED global batch of 80 split over the ranks (80 / world per GPU) -- at world = 1 the
    strong-scaling base, so that `strong.ms_per_step` at N = 1 over N = 8 is BASELINE's "strong scaling 1 -> 8" directly."""
    import torch.distributed as dist
    from tracknetv3_amd.parallel import TrackNetTrainer
    from tracknetv3_amd.utils import synth
    from tracknetv3_amd.utils.general import get_model
    in_dim = (SEQ_LEN + 1) * 3
    model = synth.init_state_(get_model("TrackNet", SEQ_LEN, BG_MODE), 31, calibrated=False).to(dev)
    from tracknetv3_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-3)          # torch.optim.Adam's arithmetic and state layout, one launch per step
    trainer = TrackNetTrainer(model, opt, alpha=0.5, seed=13)

    def barrier():
        ctl_barrier(dev)
        torch.cuda.synchronize(dev)

    def data(n, salt):
        gen = torch.Generator(device=dev).manual_seed(1234 + rank + salt)
        return (torch.rand((n, in_dim, H, W), device=dev, generator=gen),
                synth.disc_heatmaps(n, SEQ_LEN, H, W, 77 + rank + salt, device=dev))

    def timed_steps(x, y, n_steps, n_warm):
        for _ in range(n_warm):
            trainer.step(x, y)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loss = trainer.step(x, y)
        barrier()
        dt = time.perf_counter() - t0
        dt = ctl_max([dt], dev)[0]
        return dt, loss

    x, y = data(batch, 0)
    dt, loss = timed_steps(x, y, steps, warmup)
    replicas = None
    if world > 1:
        # data parallelism's invariant after K averaged updates: every rank holds the same bits.  One int64 checksum per rank over the
        # parameters' bit patterns (and one over rank 0-independent state: Adam's moments), gathered over the gloo control group.
        def checksum(tensors):
            acc = torch.zeros((), dtype=torch.int64, device=dev)
            for k, t in enumerate(tensors):
                acc += (t.detach().contiguous().view(torch.int32).to(torch.int64) * (k + 1)).sum()
            return int(acc.item())
        mine = torch.tensor([checksum(model.parameters()),
                             checksum([v for st in opt.state.values() for v in st.values() if torch.is_tensor(v) and v.dtype == torch.float32 and v.is_cuda])],
                            dtype=torch.int64)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine, group=_CTL["group"])
        replicas = {"identical": bool(all(torch.equal(c, allc[0]) for c in allc)), "ranks": world,
                    "bucket_copies": int(trainer.reducer.copies) if trainer.reducer is not None else None,
                    "note": "int64 checksums of the parameters' and Adam moments' bit patterns after the timed steps, equal on every rank; "
                            "bucket_copies: gradients that were copied into an all-reduce bucket instead of being written there by their kernel"}
    # the same step with round 5's training-forward configuration (F(4x4) on every layer, the upsampled halves in the 9-GEMM F(2x2) form): what
    # round 6's default -- first block in F(2x2), 25-of-36 upsampled halves: less heat-map error (tuning.WINO43_TRAIN_F22_LAYERS) -- costs
    alt = None
    if world == 1 and steps >= 4:
        from tracknetv3_amd import tuning as _t
        keep = (_t.WINO43_TRAIN_F22_LAYERS, _t.UP2X_WINO_VARIANT_TRAIN)
        try:
            _t.WINO43_TRAIN_F22_LAYERS, _t.UP2X_WINO_VARIANT_TRAIN = frozenset(), 0
            adt, _ = timed_steps(x, y, steps, 2)
            alt = {"ms_per_step": round(adt / steps * 1e3, 3),
                   "config": "round 5's training forward: F(4x4) on all 17 layers, upsampled halves in the 9-GEMM F(2x2) form (heat maps 4.5e-5 / "
                             "6.7e-5 / 1.1e-4 from fp64 at head gain 2.4 / 4 / 6; the default: 4.2e-5 / 5.7e-5 / 8.3e-5, profiles/r06_train_precision_sets.json)"}
        finally:
            _t.WINO43_TRAIN_F22_LAYERS, _t.UP2X_WINO_VARIANT_TRAIN = keep
    overlap = None
    if record_timing and world > 1:
        timed = TrackNetTrainer(model, opt, alpha=0.5, seed=14, record_timing=True)
        timed.step(x, y)
        barrier()
        overlap = timed.overlap_report()
    strong = None
    if strong_steps > 0 and STRONG_GLOBAL_BATCH % world == 0:
        nb = STRONG_GLOBAL_BATCH // world
        try:
            if nb == batch:
                sdt, sn = dt, steps                        # 8 ranks x 10: the weak-scaling shard IS the strong-scaling shard
            else:
                del x, y
                torch.cuda.empty_cache()
                xs, ys = data(nb, 5000)
                sdt, _ = timed_steps(xs, ys, strong_steps, 1)
                sn = strong_steps
                del xs, ys
            strong = {"global_batch": STRONG_GLOBAL_BATCH, "batch_per_gpu": nb, "n_gpus": world, "steps": sn,
                      "ms_per_step": round(sdt / sn * 1e3, 3), "value": round(STRONG_GLOBAL_BATCH * SEQ_LEN * sn / sdt, 2),
                      "unit": "frames/s", "scaling": "strong",
                      "note": "same training step at a fixed global batch of 80 (README.md:144 x 8 GPUs) split over the ranks; "
                              "N = 1 is the strong-scaling base: speed-up(N) = strong.ms_per_step(1) / strong.ms_per_step(N)"}
        except Exception as e:  # noqa: BLE001 -- e.g. out of memory on a smaller part: the weak figures must survive
            strong = {"error": f"{type(e).__name__}: {e}"}
    frames = world * batch * SEQ_LEN * steps
    ms = dt / steps * 1e3
    tf_exec = train_flops_executed_per_sample() * batch / (ms * 1e-3) / 1e12
    tf_alg = TRAIN_FLOPS_PER_SAMPLE * batch / (ms * 1e-3) / 1e12
    return {
        "metric": "frames/sec (288x512, seq_len=8) TrackNet training", "value": round(frames / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2] shard: TrackNet seq_len=8 bg_mode=concat, batch 10 per GPU, mixup alpha=0.5, "
                               "WBCE, backward, Adam(lr=1e-3) as one fused launch, mixup draws on the device; DP gradient all-reduce over RCCL when "
                               "n_gpus > 1",
                   "batch_per_gpu": batch, "global_batch": world * batch, "parallelism": f"dp{world}",
                   "rccl_world_size": (dist.get_world_size() if world > 1 else 1)},
        "roofline": {"bound": "mfma", "achieved": round(tf_exec, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf_exec / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                     "kernel": "whole step (conv fwd + dgrad + wgrad MFMA kernels + HBM-bound BN/pool/head passes)",
                     "effective_tflops": round(tf_alg, 2),
                     "note": "`achieved` / `frac`: multiply-adds the matrix pipe EXECUTES per second (whole step time, so the HBM-bound "
                             "passes count against it) over the fp32 MFMA peak; `effective_tflops` prices the same time at the "
                             "reference's algorithmic FLOP count (SURVEY 8d: 678.2 GFLOP/sample) -- the upsampled channels of the "
                             "three decoder-entry layers run at the low resolution in all three passes -- the forward in the Winograd form that keeps 9 "
                             "of the 16 F(2x2) GEMMs (9/36 of those MACs), the data and weight gradients with 25 of the 36 F(4x4) products (6.25/36) "
                             "--, the plain layers and the skip halves in fused Winograd F(4x4,3x3) form (9/36) in "
                             "all three passes (forward with the statistics epilogue, data gradient, weight gradient) --, counted per layer "
                             "from the dispatch rules (train_flops_executed_per_sample: a knob that moves a pass to F(2x2) or the direct form "
                             "moves its count to 16/36 or 1)"},
        "strong": strong, "dp_overlap": overlap, "replicas": replicas, "round5_forward_config": alt, "final_loss": round(float(loss.item()), 6)}


def bench_train(args, dev, rank, world):
    import torch.distributed as dist
    rccl = _CTL["rccl"]
    if rccl is not None and not rccl["ok"]:
        if rank == 0:
            print(json.dumps({"metric": "frames/sec (288x512, seq_len=8) TrackNet training", "value": None, "n_gpus": world,
                              "error": "RCCL first contact failed: " + str(rccl.get("error")), "rccl": rccl}), flush=True)
        sys.stdout.flush()
        os._exit(3)
    out = train_leg(dev, rank, world, args.batch, args.steps, args.warmup, record_timing=True, strong_steps=args.strong_steps)
    out["rccl"] = rccl
    if rank == 0:
        out["cpu_baseline"] = cpu_baseline("train", budget_s=30.0) if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---- control plane of an N > 1 run.  The bench's OWN barriers and reductions of timings go through a gloo group (host tensors): they are
#      measurement plumbing and must not depend on the transport under test.  RCCL is used where the product uses it -- the gradient
#      all-reduce of the training step (tracknetv3_amd/parallel.py) -- after one watched first contact: if communicator set-up or the
#      first all-reduce does not finish in RCCL_WATCHDOG_S seconds, every rank reports who hung (rank, device, NCCL_DEBUG tail), rank 0 prints
#      an {"error": ...} line and all ranks leave through os._exit at once -- WITHOUT synchronising the device, on which the hung collective's
#      kernel may still spin (a later torch.cuda.synchronize would block until the process-group timeout).  A first contact that FAILS
#      without hanging (an exception, a wrong sum) only turns the training leg into {"error": ...}; the inference leg (no collective in its
#      data path) is still measured.  HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) is set for every
#      rank: this host driver has no legacy IPC handles, and RCCL's intra-node transport fails without it (hipIpcGetMemHandle: invalid argument).
RCCL_WATCHDOG_S = float(os.environ.get("TNV3_RCCL_WATCHDOG_S", "60"))
_CTL = {"group": None, "rccl": None}


def _dev_sync(dev):
    if getattr(dev, "type", "cuda") == "cuda":
        torch.cuda.synchronize(dev)


def ctl_barrier(dev):
    _dev_sync(dev)
    if _CTL["group"] is not None:
        import torch.distributed as dist
        dist.barrier(group=_CTL["group"])


def ctl_max(vals, dev):
    if _CTL["group"] is None:
        return [float(v) for v in vals]
    import torch.distributed as dist
    t = torch.tensor(list(vals), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CTL["group"])
    return [float(v) for v in t.tolist()]


def _nccl_debug_tail(rank, lines=12):
    path = os.environ.get("NCCL_DEBUG_FILE", "").replace("%r", str(rank)).replace("%h", "host").replace("%p", str(os.getpid()))
    try:
        with open(path) as f:
            return f.read().splitlines()[-lines:]
    except OSError:
        return []


def rccl_first_contact(dev, rank, world, backend):
    """One all-reduce over the default (RCCL) group under a watchdog.  Returns {"ok": True, "first_allreduce_ms": ...} or
    {"ok": False, "error": ..., "rank": ..., "nccl_debug_tail": [...]}; every rank learns the verdict of all (gloo MIN)."""
    import threading
    import torch.distributed as dist
    res = {}

    def probe():
        try:
            t = torch.ones(1 << 20, device=dev if backend == "nccl" else "cpu")      # (gloo: the --share-gpus test mode, host tensors)
            _dev_sync(dev)
            t0 = time.perf_counter()
            dist.all_reduce(t)
            _dev_sync(dev)
            res["ms"] = (time.perf_counter() - t0) * 1e3
            res["ok"] = bool(abs(float(t[0].item()) - world) < 1e-3)
            if not res["ok"]:
                res["error"] = f"all-reduce of ones over {world} ranks returned {float(t[0].item())}"
        except Exception as e:  # noqa: BLE001
            res["ok"], res["error"] = False, f"{type(e).__name__}: {e}"

    th = threading.Thread(target=probe, daemon=True)
    th.start()
    th.join(RCCL_WATCHDOG_S)
    if th.is_alive():
        res = {"ok": False, "error": f"no answer from the first {backend} all-reduce within {RCCL_WATCHDOG_S:.0f} s (communicator set-up or the collective hangs)",
               "hung": True}
    mine = 1.0 if res.get("ok") else 0.0
    flags = torch.tensor([mine, 0.0 if res.get("hung") else 1.0], dtype=torch.float64)      # (ok, not hung) -- MIN over ranks, over the gloo control group
    dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=_CTL["group"])
    out = {"ok": bool(flags[0].item() > 0.5), "backend": backend, "watchdog_s": RCCL_WATCHDOG_S, "any_hung": bool(flags[1].item() < 0.5)}
    if res.get("ok"):
        out["first_allreduce_ms"] = round(res["ms"], 2)
    if not out["ok"]:
        out["error"] = res.get("error", "another rank failed its first contact")
        out["rank"], out["device"], out["hung_here"] = rank, str(dev), bool(res.get("hung"))
        out["nccl_debug_tail"] = _nccl_debug_tail(rank)
        print(f"[bench] rank {rank} on {dev}: RCCL first contact FAILED: {out['error']}", file=sys.stderr, flush=True)
        for line in out["nccl_debug_tail"]:
            print(f"[bench] rank {rank} NCCL: {line}", file=sys.stderr, flush=True)
    return out


def self_spawn(n, share_gpus=False):
    """`python bench.py --gpus N` with no launcher around it: re-execute under torch.distributed.run, one rank per GPU (RCCL).
    Refuses (non-zero exit) when the node has fewer than N GPUs -- never a silent 1-rank run.  (share_gpus: the launcher TEST mode,
    see --share-gpus.)"""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if share_gpus else n):
        raise SystemExit(f"bench.py --gpus {n} needs {n} GPUs on this node, found {have}: refusing to run with fewer ranks")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("NCCL_DEBUG", "WARN")                   # what a failed first contact prints (rccl_first_contact reads the tail)
    env.setdefault("NCCL_DEBUG_FILE", os.path.join(os.environ.get("TMPDIR", "/tmp"), "tnv3_bench_nccl_%r.log"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] spawning: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _event_ms(fn, reps, dev, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def inpaintnet_leg(dev):
    """BASELINE configs[3]: InpaintNet, sequence length 16 -- forward latency at the README batch of 32 (README.md:162), forward
    throughput at 65 536 windows, and the train step of train.py:147-166 (mask, forward, masked MSE, backward,
    clip_grad_norm_(1), Adam) at batch 32."""
    from tracknetv3_amd.optim import FusedAdam
    from tracknetv3_amd.utils.general import get_model
    net = get_model("InpaintNet").to(dev).eval()
    out = {"workload": "BASELINE configs[3]: InpaintNet seq_len=16 on (x, y, vis) sequences, random-init weights"}
    for n, reps, key in ((32, 200, "fwd_n32"), (65536, 10, "fwd_n65536")):
        x = torch.rand(n, 16, 2, device=dev)
        m = (torch.rand(n, 16, 1, device=dev) < 0.3).float()
        with torch.no_grad():
            ms = _event_ms(lambda: net(x, m), reps, dev)
        out[key] = {"ms": round(ms, 4), "seq_per_s": round(n / ms * 1e3, 1), "tflops": round(16.63e6 * n / ms / 1e9, 3)}
    net.train()
    opt = FusedAdam(net.parameters(), lr=1e-3, max_grad_norm=1.0)
    n = 32
    coor = torch.rand(n, 16, 2, device=dev)
    gt = torch.rand(n, 16, 2, device=dev)
    mask = (torch.rand(n, 16, 1, device=dev) < 0.3).float()
    mse = torch.nn.MSELoss()

    def step():
        opt.zero_grad(set_to_none=True)
        o = net(coor * (1 - mask), mask)
        mse(o * mask, gt * mask).backward()
        opt.step()

    ms = _event_ms(step, 50, dev, warm=3)
    out["train_n32"] = {"ms_per_step": round(ms, 4), "seq_per_s": round(n / ms * 1e3, 1),
                        "note": "train.py:147-166 at the README batch: forward + masked MSE + backward + clip_grad_norm_(1) + Adam"}
    return out


def e2e_leg(dev, t_frames=256):
    """BASELINE configs[4]: predict.py's flow on a synthetic 1080p uint8 stream resident in HBM -- temporal median, Pillow-exact
    bicubic resize to 288x512, TrackNet(8, concat) windows, temporal ensemble, peak-find, InpaintNet(16), final coordinates.
    FPS = source frames per second, per eval_mode (`nonoverlap`: one network pass per 8 frames; `weight`: one per frame)."""
    from tracknetv3_amd.pipeline import predict_video
    from tracknetv3_amd.utils import synth
    from tracknetv3_amd.utils.general import get_model
    tn = synth.init_state_(get_model("TrackNet", SEQ_LEN, BG_MODE), 31, calibrated=True).to(dev).eval()
    net = get_model("InpaintNet").to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(99)
    bg = torch.randint(0, 96, (1, 1080, 1920, 3), dtype=torch.uint8, device=dev, generator=gen)
    src = bg.repeat(t_frames, 1, 1, 1)
    for f in range(t_frames):                     # a bright 17-px square moving over a static textured background
        cx, cy = 100 + 6 * (f % 280), 300 + (f * 7) % 500
        src[f, cy - 8:cy + 9, cx - 8:cx + 9] = 255
    out = {"workload": "BASELINE configs[4]: predict.py flow on a synthetic 1080p uint8 stream (median + bicubic resize + "
                       "TrackNet(8,concat) + ensemble + peak-find + InpaintNet(16)), batch 16, frames resident in HBM as uint8",
           "frames": t_frames, "readme_fps": 25.11, "readme_note": "README.md:31, hardware unstated, eval_mode weight"}
    for mode in ("nonoverlap", "weight"):
        predict_video(src, tn, net, SEQ_LEN, 16, BG_MODE, mode, 16)      # warm-up: allocator pools of both streams
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(3):                         # (median of three: one slow run of two moved the 256-frame `weight` figure by 30 % once)
            t0 = time.perf_counter()
            pd = predict_video(src, tn, net, SEQ_LEN, 16, BG_MODE, mode, 16)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        assert len(pd["Frame"]) == t_frames
        dt = float(np.median(ts))
        out[mode] = {"fps": round(t_frames / dt, 1), "s": round(dt, 4), "runs": len(ts), "visible": int(sum(pd["Visibility"])),
                     "vs_readme": round(t_frames / dt / 25.11, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE, else 1); > 1 without a launcher: bench.py spawns the ranks")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=10, help="windows per GPU per step (BASELINE configs[1]: 10)")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="infer: BASELINE configs[1] (headline); train: configs[2] shard -- mixup + fwd + WBCE + bwd + Adam")
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=3,
                    help="the K-step headline block is timed this many times (each exactly K steps between barriers); `value` is the "
                         "median block, all of them are listed")
    ap.add_argument("--train-steps", type=int, default=5,
                    help="infer mode: also time this many configs[2]-shard training steps (2 warm-ups) and report them as `train` (0 = skip)")
    ap.add_argument("--strong-steps", type=int, default=3,
                    help="training leg: also time this many steps at the fixed global batch of 80 split over the ranks (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, choices=(0, 1),
                    help="N = 1 only: append the `inpaintnet` (configs[3]) and `e2e` (configs[4]) sub-objects")
    ap.add_argument("--infer-split", type=int, default=None, choices=(0, 1),
                    help="force the intra-batch two-stream split of the eval forward off / on (default: the product's setting)")
    ap.add_argument("--overlap-streams", type=int, default=2,
                    help="also time the K steps round-robin on this many HIP streams (reported as `overlap`; 0/1 = skip)")
    ap.add_argument("--layers-out", default=os.path.join(ROOT, "gpurun_out", "bench_layers.json"))
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend of the N > 1 run: nccl = RCCL over xGMI (the product path); gloo only for --share-gpus")
    ap.add_argument("--share-gpus", action="store_true",
                    help="TEST mode for boxes with fewer GPUs than ranks: rank r uses GPU r %% device_count and the ranks talk over gloo "
                         "(RCCL refuses two ranks on one device).  Exercises the launcher, the rendezvous, the max-over-ranks timing and "
                         "the data-parallel step end to end; its numbers are NOT a scaling measurement and the line says so")
    args = ap.parse_args()
    if args.share_gpus:
        args.backend = "gloo"
        # TEST MODE only: more than two processes sharing one device oversubscribe its hardware queues (per process: main, weight-gradient,
        # reducer and gloo's copy streams), and with the queues time-sliced this image loses stores of kernels in flight / aborts with an HSA
        # illegal-instruction error at ~0.13 % of the rank-steps (7 of 5 544) (reproducer scripts/dp_soak.py; 0 faults in 8192 rank-steps
        # with two hardware queues per process).  So the ranks of this mode run the PRODUCT's stream topology on two hardware queues each
        # (HIP multiplexes the streams onto them); one process per GPU -- every real run -- keeps the runtime's default.
        if args.gpus and args.gpus > 2:
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")      # read when the HIP runtime initialises: before the first device call; inherited by spawned ranks
    elif args.backend == "gloo":
        raise SystemExit("--backend gloo is only meaningful with --share-gpus (the product's collective path is RCCL)")

    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not launched:
        sys.exit(self_spawn(args.gpus, args.share_gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the rank count must equal --gpus")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs {args.gpus} GPU(s), found 0 (there is no CPU fallback for the product path)")
    if torch.cuda.device_count() < world and not args.share_gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {torch.cuda.device_count()}")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if args.share_gpus else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=max(120.0, 4 * RCCL_WATCHDOG_S)))
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        _CTL["group"] = dist.new_group(backend="gloo")       # the bench's own barriers / reductions: never the transport under test
        _CTL["rccl"] = rccl_first_contact(dev, rank, world, args.backend)
        if not _CTL["rccl"]["ok"] and _CTL["rccl"].get("any_hung"):
            # A collective that HANGS may have left a kernel spinning on this rank's device: every later torch.cuda.synchronize(dev) -- the
            # barriers of the inference leg -- would block behind it until the process-group timeout kills the job.  Report now and leave:
            # no device synchronisation, no teardown (a thread still sits inside the communicator).
            if rank == 0:
                print(json.dumps({"metric": "frames/sec (288x512, seq_len=8) TrackNet inference", "value": None, "unit": "frames/s", "n_gpus": world,
                                  "error": "RCCL first contact hung: " + str(_CTL["rccl"].get("error")), "rccl": _CTL["rccl"],
                                  "note": "nothing was measured: a hung collective can hold the device, so the ranks leave without synchronising it; "
                                          "`python bench.py --gpus 1` measures the single-GPU legs"}), flush=True)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)
    n_gpus = world

    from tracknetv3_amd import _lib, ops
    from tracknetv3_amd.utils import synth
    from tracknetv3_amd.utils.general import get_model
    _lib.load()
    assert not _lib.is_emulator()

    if args.tune and rank == 0:
        tune(dev, args.batch, [os.path.join(ROOT, "gpurun_out", "conv_tuning.json")])
    ctl_barrier(dev)

    if args.mode == "train":
        return bench_train(args, dev, rank, world)

    in_dim = (SEQ_LEN + 1) * 3
    model = synth.init_state_(get_model("TrackNet", SEQ_LEN, BG_MODE), 31, calibrated=True).to(dev).eval()
    x = torch.rand((args.batch, in_dim, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))

    # per-launch timing of the dominant kernel family: HIP events on the launch stream
    layers = conv_layer_table(in_dim, H, W)
    events = []
    ops_conv, ops_up2x, ops_wino, ops_up2xw, ops_w43 = ops.conv3x3, ops.conv_up2x, ops.conv3x3_wino, ops.conv_up2x_wino, ops.conv3x3_wino43

    def timed(kind, fn):
        def wrap(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            events.append((kind, e0, e1))
            return out
        return wrap

    timed_conv, timed_up2x, timed_wino = timed("conv", ops_conv), timed("up2x", ops_up2x), timed("conv", ops_wino)
    timed_up2xw, timed_w43 = timed("up2x", ops_up2xw), timed("conv", ops_w43)

    def barrier():
        ctl_barrier(dev)
        torch.cuda.synchronize(dev)

    def max_over_ranks(vals):
        return ctl_max(vals, dev)

    def block(n_steps):
        """Exactly n_steps forward passes between barrier + synchronize on both sides; seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            model(x)
        barrier()
        return max_over_ranks([time.perf_counter() - t0])[0]

    from tracknetv3_amd import tuning as _tuning
    from tracknetv3_amd.model import no_infer_split
    if args.infer_split is not None:
        _tuning.INFER_SPLIT = bool(args.infer_split)
    split_on = bool(_tuning.INFER_SPLIT and args.batch >= _tuning.INFER_SPLIT_MIN_BATCH)
    n_blocks = max(1, args.blocks)

    # the headline: K steps of the product's default path (the batch split 6 : 4 over two HIP streams unless switched off),
    # timed n_blocks times; `value` is the median block
    for _ in range(args.warmup):
        model(x)
    blocks_s = [block(args.steps) for _ in range(n_blocks)] if split_on else None

    # the roofline pass: the same K steps with the whole batch on ONE stream and HIP events around every conv launch, so that a
    # launch's duration is its own (co-running launches of the other half would stretch each other's event brackets).  With the
    # split off (--infer-split 0: the profiling scripts) this IS the headline pass.
    with no_infer_split():
        if split_on:
            for _ in range(min(args.warmup, 2)):
                model(x)
        ops.conv3x3, ops.conv_up2x, ops.conv3x3_wino, ops.conv_up2x_wino, ops.conv3x3_wino43 = timed_conv, timed_up2x, timed_wino, timed_up2xw, timed_w43
        single_s = [block(args.steps) for _ in range(n_blocks)]
        ops.conv3x3, ops.conv_up2x, ops.conv3x3_wino, ops.conv_up2x_wino, ops.conv3x3_wino43 = ops_conv, ops_up2x, ops_wino, ops_up2xw, ops_w43
    if not split_on:
        blocks_s = single_s
    dt = float(np.median(blocks_s))
    dt_single = float(np.median(single_s))

    # Extra (reported beside, never instead of, `value`): the same K steps on independent batches issued round-robin on two
    # HIP streams.  Windows are independent, so the second stream's launches fill the CUs that the 45/48 tail of every
    # batch-10 conv launch leaves idle (profiles/r01_conv_batch32_vs_batch10.md); pipeline.predict_video runs this way.
    dt2 = None
    if args.overlap_streams > 1:
        side = [torch.cuda.Stream(dev) for _ in range(args.overlap_streams)]
        xs = [x] + [torch.rand_like(x) for _ in range(args.overlap_streams - 1)]
        model.prepare_eval()
        barrier()
        with no_infer_split():                              # whole batches in flight, as pipeline.predict_video keeps them
            for k in range(2 * len(side)):
                with torch.cuda.stream(side[k % len(side)]):
                    model(xs[k % len(side)])
            barrier()
            t0 = time.perf_counter()
            for k in range(args.steps):
                with torch.cuda.stream(side[k % len(side)]):
                    model(xs[k % len(side)])
            barrier()
            dt2 = max_over_ranks([time.perf_counter() - t0])[0]
        del xs, side

    if rank == 0:
        # A decoder-entry layer is two launches: conv_up2x (its upsampled channels, at the low resolution) followed by
        # the conv3x3 of its skip channels that takes the partial sums as addend; both count towards that layer.
        # Per layer: the MEDIAN over all timed steps (one stall inside one event bracket -- allocator miss, a sampler on the
        # box -- must not move the roofline figure); min / max are kept in the layers file.
        total_steps = args.steps * n_blocks
        per = np.zeros((total_steps, 17))
        n_launch, k, carry = 0, 0, 0.0
        for kind, e0, e1 in events:
            n_launch += 1
            if kind == "up2x":
                carry += e0.elapsed_time(e1)
                continue
            per[k // 17, k % 17] = e0.elapsed_time(e1) + carry
            carry, k = 0.0, k + 1
        assert k == 17 * total_steps, (k, len(events))
        per_layer_ms = np.median(per, axis=0)
        per_layer_min, per_layer_max, per_layer_mean = per.min(axis=0), per.max(axis=0), per.mean(axis=0)
        launches_per_step = n_launch // total_steps
        fl = np.array([conv_flops(c0, c1, co, h, w) * args.batch for (_, c0, c1, co, h, w, _) in layers])
        # multiply-adds the device executes: the upsampled channels (c0 of an `up` layer) cost 4 taps instead of 9, and the
        # plain halves that run in Winograd F(2x2,3x3) form cost 16 instead of 36 per 2x2 tile
        # (Winograd F(4x4,3x3), where the eval forward uses it: 36 per 4x4 tile = 9/36 of the direct count)
        def plain_frac(ci, co, h, w, skip_half=False):
            if (not skip_half or ci >= _tuning.WINOGRAD_MIN_SKIP) and _tuning.use_wino43(ci, co, h, w):
                return 9 / 36
            return 16 / 36 if _tuning.use_winograd(ci, co, h, w) else 1.0

        def executed(c0, c1, co, h, w, up):
            if up:
                skip = conv_flops(c1, 0, co, h, w) * plain_frac(c1, co, h, w, skip_half=True)
                uv = ops.up2x_wino_variant(_tuning.UP2X_WINO_VARIANT)
                if _tuning.UP2X_WINO and ops.up2x_wino_supported(c0, co, h // 2, w // 2, uv):
                    up_frac = (6.25 if uv == 2 else 9) / 36          # 25 of the 36 F(4x4) products per 4x4 tile / 9 of the 16 F(2x2) GEMMs
                elif _tuning.UP2X_WINO and ops.up2x_wino_supported(c0, co, h // 2, w // 2, 0):
                    up_frac = 9 / 36
                else:
                    up_frac = 4 / 9
                return conv_flops(c0, 0, co, h, w) * up_frac + skip
            return conv_flops(c0, c1, co, h, w) * plain_frac(c0, co, h, w)
        fl_exec = np.array([executed(c0, c1, co, h, w, up) * args.batch for (_, c0, c1, co, h, w, up) in layers])
        conv_ms = float(per_layer_ms.sum())
        effective = float(fl.sum() / conv_ms / 1e9)              # the reference's algorithmic FLOPs per second of conv-kernel time
        achieved = float(fl_exec.sum() / conv_ms / 1e9)          # what the matrix pipe really executes per second
        frames = n_gpus * args.batch * SEQ_LEN * args.steps
        ms_per_step = dt / args.steps * 1e3
        layer_rows = [{"layer": layers[k][0], "ms": round(float(per_layer_ms[k]), 4), "min_ms": round(float(per_layer_min[k]), 4),
                       "max_ms": round(float(per_layer_max[k]), 4), "mean_ms": round(float(per_layer_mean[k]), 4),
                       "tflops": round(float(fl[k] / per_layer_ms[k] / 1e9), 2),
                       "executed_tflops": round(float(fl_exec[k] / per_layer_ms[k] / 1e9), 2)} for k in range(17)]
        try:
            os.makedirs(os.path.dirname(args.layers_out), exist_ok=True)
            with open(args.layers_out, "w") as f:
                json.dump({"ms_per_step": ms_per_step, "conv_ms_per_step": conv_ms, "steps_timed": total_steps,
                           "statistic": "median over the timed steps (min / max / mean beside it)", "layers": layer_rows}, f, indent=1)
        except OSError:
            pass
        for r in layer_rows:
            print(f"[layer] {r['layer']:22s} {r['ms']:8.3f} ms (min {r['min_ms']:.3f} max {r['max_ms']:.3f})  {r['tflops']:7.2f} TFLOP/s  "
                  f"executed {r['executed_tflops']:6.2f}", file=sys.stderr)
        # HBM bytes per conv launch come from separate rocprofv3 --pmc passes over THIS command (rocprofv3 cannot wrap itself;
        # scripts/gpu_session.sh `pmc` + scripts/conv_traffic.py write profiles/conv_traffic.json with the commit and time of
        # the passes).  Reported only when that file describes the same launch list; otherwise null, never a stale replay.
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
                tj = json.load(f)
            sha, ssha = lib_sha256(), src_sha256()
            # the counters belong to a BUILD: the same kernel sources + flags (path-independent), or -- for counter files
            # written before that key existed -- the same library file
            same_build = (ssha is not None and tj.get("src_sha256") == ssha) or (tj.get("src_sha256") is None and sha is not None and tj.get("lib_sha256") == sha)
            if int(tj.get("conv_launches_per_step", -1)) == launches_per_step and tj.get("kernel_set") == KERNEL_SET and same_build:
                traffic = float(tj["traffic_bytes_per_launch"])
                traffic_src = {"file": "profiles/conv_traffic.json", "commit": tj.get("commit"), "taken_utc": tj.get("taken_utc"), "lib_sha": sha,
                               "src_sha": ssha,
                               "note": "PMC FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE over the conv launches of this same command, taken on "
                                       "a build of the same kernel sources and flags (src_sha256); any other build reports null"}
        except (OSError, KeyError, ValueError, TypeError):
            pass
        to_ms = lambda sec: round(sec / args.steps * 1e3, 4)      # noqa: E731
        out = {
            "metric": "frames/sec (288x512, seq_len=8) TrackNet inference", "value": round(frames / dt, 2), "unit": "frames/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "blocks": {"count": n_blocks, "steps_each": args.steps, "ms_per_step": [to_ms(b) for b in blocks_s],
                       "min": to_ms(min(blocks_s)), "median": to_ms(dt), "max": to_ms(max(blocks_s)),
                       "note": "each block = exactly K steps between barrier + synchronize, max over ranks; `value` / `ms_per_step` "
                               "are the median block"},
            "config": {"workload": "BASELINE configs[1]: TrackNet seq_len=8 bg_mode=concat eval forward, batch 10 per GPU, "
                                   "288x512 synthetic frames, synthetic (PRNG) weights", "batch_per_gpu": args.batch,
                       "frames_per_step": n_gpus * args.batch * SEQ_LEN, "parallelism": f"replicated windows x{n_gpus} (no collective)",
                       "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
                       "backend": (args.backend if world > 1 else None),
                       "shared_gpus": (f"TEST MODE: {world} ranks on {torch.cuda.device_count()} GPU(s) over gloo -- not a scaling measurement"
                                       + (f"; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} per process (hardware-queue oversubscription: DESIGN 5e)" if world > 2 else "")
                                       if args.share_gpus else None),
                       "launcher": ("torch.distributed.run" if launched else "single process"),
                       "schedule": ("the batch's images split 6 : 4 over two HIP streams (product default, tuning.INFER_SPLIT): the halves' "
                                    "per-layer launches overlap at their tails; outputs bit-identical to the one-stream forward")
                       if split_on else "whole batch on one HIP stream"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "bytes per launch",
                         "kernel": f"conv3x3_wino43s_kernel<CBW 4 | 8, MODE 0 plain | 1 upsampled half> ({launches_per_step} launches/step for the 17 "
                                   f"conv layers, fp32 MFMA 16x16x4)",
                         "lib_sha256": lib_sha256(), "src_sha256": src_sha256(),
                         "measured": "second pass of the same K-step blocks with the whole batch on ONE stream (model.no_infer_split) and HIP "
                                     "events around every conv launch: a launch's duration is its own, not stretched by the other half's "
                                     "co-running launches; per layer the MEDIAN over all timed steps; `single_stream_ms_per_step` is "
                                     "that pass's median block",
                         "single_stream_ms_per_step": to_ms(dt_single),
                         "single_stream_blocks_ms_per_step": [to_ms(b) for b in single_s],
                         "step_view_tflops": round(float(fl_exec.sum()) / (dt / args.steps) / 1e12, 2),
                         "avg_launch_ms": round(conv_ms / launches_per_step, 4), "conv_ms_per_step": round(conv_ms, 4),
                         "conv_ms_per_step_mean": round(float(per_layer_mean.sum()), 4),
                         "algorithmic_gflop_per_step": round(float(fl.sum()) / 1e9, 3),
                         "executed_gflop_per_step": round(float(fl_exec.sum()) / 1e9, 3),
                         "effective_tflops": round(effective, 2),
                         "speedup_vs_direct_flops": round(float(fl.sum() / fl_exec.sum()), 3),
                         "note": "`achieved` / `frac` = FLOPs the matrix pipe EXECUTES per second over the fp32 MFMA peak (an honest "
                                 "roofline position, <= 1).  Every 3x3 layer of the eval forward runs in fused Winograd F(4x4,3x3) form on the "
                                 "16x16x4 kernel: the 14 plain layers and the 3 skip halves with 36 products per 4x4 output tile (9/36 of the "
                                 "direct multiply-adds), the 3 upsampled halves of the decoder entries at the low resolution with 25 of the 36 "
                                 "(6.25/36), so the executed count is `executed_gflop_per_step`; "
                                 "`effective_tflops` prices the same kernel time at the reference's algorithmic count "
                                 "(2*9*Cin*Cout*H*W per layer, SURVEY 8d) and may exceed the peak -- it is a speed-up over the direct "
                                 "form, not a roofline fraction",
                         "hbm_view": {"algorithmic_GB_per_step": round(ALG_BYTES_PER_SAMPLE * args.batch / 1e9, 3),
                                      "achieved_GBps": round(ALG_BYTES_PER_SAMPLE * args.batch / (ms_per_step * 1e-3) / 1e9, 1),
                                      "peak_GBps": PEAK_HBM_GBPS}},
        }
        out["overlap"] = None if dt2 is None else {
            "streams": args.overlap_streams, "value": round(frames / dt2, 2), "unit": "frames/s",
            "ms_per_step": round(dt2 / args.steps * 1e3, 4),
            "note": "same K steps, independent batches round-robin on HIP streams; not used for `value` or `roofline`"}
    # The other half of BASELINE's metric ("train+infer"): the configs[2] shard, a few steps, in the same driver-timed record.
    train = None
    del model, x
    torch.cuda.empty_cache()
    rccl = _CTL["rccl"]
    if args.train_steps > 0:
        if rccl is not None and not rccl["ok"]:            # the collective never came up: the inference line above stands, the training leg says why
            train = {"error": "RCCL first contact failed: " + str(rccl.get("error"))}
        else:
            try:
                train = train_leg(dev, rank, world, args.batch, args.train_steps, 2, record_timing=True, strong_steps=args.strong_steps)
            except Exception as e:  # noqa: BLE001 -- the inference line must survive a failing training leg (e.g. a collective error)
                train = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if rank == 0:
        out["train"] = train
        out["rccl"] = rccl
        if args.extras and n_gpus == 1:
            # (e2e: the 256-frame clip of rounds 2-5 -- two TrackNet batches of 16 windows: a fifth of its time is pipeline fill and drain -- and, since
            #  round 6, a 1024-frame clip beside it: the same flow where the batches overlap one another's pre- and post-processing)
            for key, leg in (("inpaintnet", inpaintnet_leg), ("e2e", e2e_leg), ("e2e_1024_frames", lambda d: e2e_leg(d, 1024))):
                try:
                    out[key] = leg(dev)
                except Exception as e:  # noqa: BLE001 -- secondary legs never cost the headline line
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline("infer")
            if isinstance(train, dict) and "error" not in train:
                train["cpu_baseline"] = cpu_baseline("train", budget_s=25.0)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        ctl_barrier(dev)
        if rccl is not None and not rccl["ok"]:
            sys.stdout.flush()
            os._exit(3)                                    # a thread may still sit inside the communicator: no clean teardown; non-zero for the launcher
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
