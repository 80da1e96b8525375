"""Test tooling: the fp64 / fp32 host-oracle evaluations of the full-size GPU parity tests, run in a few worker PROCESSES side by side.

One 288x512 oracle evaluation is memory-bound and stops scaling at ~32 threads (bench.py's thread scan: 16 -> 3.3 s, 32 -> 3.3 s, 64 -> 4.9 s
per batch-2 training step), while the GPU box has 128 cores: four workers x 32 threads cut the wall time of a sweep of independent
evaluations ~3x.  Workers are SPAWNED (never forked from a process that holds a HIP context) and only ever touch the CPU.
"""
import os
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp


def _threads(workers):
    cores = max(2, (os.cpu_count() or 2) // 2)          # physical cores (SMT threads only slow fp64 convolutions down)
    return max(1, min(32, cores // max(1, workers)))


def _job(spec):
    """spec: dict(kind='train'|'eval', in_dim, out_dim, seed, gain, var_range, n, h, w, dtype='float64'|'float32', threads).
    train -> (loss, heat maps, {name: gradient}, new BN buffers); eval -> logits."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import torch
    from oracle import nets
    torch.set_num_threads(int(spec["threads"]))
    dt = torch.float64 if spec["dtype"] == "float64" else torch.float32
    sd = nets.synth_state(nets.tracknet_state_shapes(spec["in_dim"], spec["out_dim"]), spec["seed"], calibrated=True, gain=spec.get("gain"),
                          var_range=spec.get("var_range"))
    x = nets.synth_input((spec["n"], spec["in_dim"], spec["h"], spec["w"]), spec["seed"] + 1000)
    if spec["kind"] == "eval":
        with torch.no_grad():
            sdd = {k: (v.to(dt) if v.dtype != torch.int64 else v) for k, v in sd.items()}
            return nets.tracknet_forward(sdd, x.to(dt), training=False, return_logits=True)
    y = nets.disc_heatmaps(spec["n"], spec["out_dim"], spec["h"], spec["w"], spec["seed"] + 2000)
    loss, p, grads, stats = nets.tracknet_train_step_grads(sd, x, y, dt)
    return loss, p, dict(grads), dict(stats)


def run(specs, workers=4):
    """Evaluate the specs (order kept).  workers <= 1 or a single spec: in this process."""
    specs = list(specs)
    workers = max(1, min(int(workers), len(specs), max(1, (os.cpu_count() or 2) // 16)))
    for s in specs:
        s.setdefault("threads", _threads(workers))
    if workers <= 1:
        return [_job(s) for s in specs]
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
        return list(pool.map(_job, specs))
