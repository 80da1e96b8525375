"""GPU (-m gpu), LAST file of the suite on purpose: BASELINE configs[2]'s rank count -- eight -- rehearsed on a box with ONE MI355X.

What round 5 ran into here (one rank's loss 2e-4 off in 1 of 7 eight-rank steps, an `HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION ... code: 0x2a`
abort in 2 of 9 rehearsals) was taken apart in round 6 (reproducers scripts/dp_soak.py, dp_soak_stock.py, forward_soak.py,
idle_first_kernel_probe.py; tables profiles/r06_eight_process_fault.json):
  * it needs eight processes SHARING the device with the full data-parallel stream set (main, weight-gradient, reducer and gloo's copy
    streams: ~40 HIP hardware queues for the device to time-slice); forward-only (63 241 rank-steps), single-kernel-after-idle
    (60 336) and gloo-free (45 600) eight-process soaks are clean, and so is the same topology running stock kernels only (4 112);
  * the damage is always the same: whole output tiles of the FIRST kernel a rank launches after waiting for the others still hold what
    that memory held before (the workgroup's early stores are missing, its later ones arrived), at 0.13 % of the rank-steps (7 of 5 544), plus the
    occasional HSA abort; re-running the layer at once gives the right bits;
  * with two hardware queues per process (GPU_MAX_HW_QUEUES=2: HIP multiplexes the same streams) it does not happen: 0 of 8 192 rank-steps.
So the ranks of these tests run the PRODUCT's stream topology (side-stream weight gradients writing into the all-reduce buckets) on two
hardware queues each (tests/test_gpu_dp.py::_worker, bench.py --share-gpus).  A NUMERIC deviation fails the test at once -- no retry
(ADVICE r5); only a process that dies (the HSA abort) is retried, once, and reported.  One process per GPU -- every deployment, every other
test -- never oversubscribes the queues and keeps the runtime's default.
"""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import nets
from test_gpu_dp import _free_port, _worker

pytestmark = pytest.mark.gpu

ATTEMPTS = 2            # for a rank that DIES (HSA abort under queue oversubscription); a wrong number is never retried


def _note(name, attempts):
    from conftest import write_report
    if any(not a["ok"] for a in attempts):
        warnings.warn(f"{name}: {sum(not a['ok'] for a in attempts)} of {len(attempts)} attempts failed under eight-process oversubscription: "
                      + json.dumps([a for a in attempts if not a["ok"]])[:1500])
    write_report(f"{name}_attempts.json", {"attempts": attempts})


def test_eight_rank_train_step_equals_sequential_eight_shard_oracle(gpu_device):
    """Eight ranks share cuda:0 over gloo (RCCL refuses two ranks on one device; on an 8-GPU node the same calls run over RCCL), global batch
    16 at 64x128, two samples per rank (with ONE, the deepest BatchNorm layers see 128 values per channel and any two fp32 evaluations differ
    by 7e-2 of max|g| on single tensors).  Must equal the DP definition: the oracle runs the eight shards one after the other (local
    BatchNorm), averages the gradient sets, applies one SGD step.  Also: every gradient is written by its kernel into the all-reduce bucket
    (copies == 0), and the replicas end bit-identical."""
    world, batch = 8, 16
    sd = nets.synth_state(nets.tracknet_state_shapes(9, 3), 13, calibrated=True)
    x = nets.synth_input((batch, 9, 64, 128), 1013)
    y = nets.disc_heatmaps(batch, 3, 64, 128, 2013)
    g64, g32, l64, st64 = [], [], [], []
    for r in range(world):
        l, _, g, st = nets.tracknet_train_step_grads(sd, x[2 * r:2 * r + 2], y[2 * r:2 * r + 2], torch.float64)
        _, _, gf, _ = nets.tracknet_train_step_grads(sd, x[2 * r:2 * r + 2], y[2 * r:2 * r + 2], torch.float32)
        g64.append(g); g32.append(gf); l64.append(l.item()); st64.append(st)

    def check(res):
        for r in range(world):
            assert abs(res[r]["loss"] - l64[r]) <= 2e-5, (r, l64[r], [res[q]["loss"] for q in range(world)])
            for k, v in res[r]["bn"].items():                   # BatchNorm running statistics stay local: rank r holds shard r's
                assert torch.allclose(v.double(), st64[r][k], rtol=2e-4, atol=2e-6), (r, k)
        mine, ref = [], []
        for name in g64[0]:
            avg64 = sum(g[name] for g in g64) / world
            avg32 = sum(g[name].double() for g in g32) / world
            for r in range(1, world):
                assert torch.equal(res[0]["grads"][name], res[r]["grads"][name]), f"rank {r} holds a different averaged gradient for {name}"
                assert torch.equal(res[0]["params"][name], res[r]["params"][name]), f"replica {r} diverged on {name}"
            assert torch.equal(res[0]["params"][name], sd[name] - res[0]["grads"][name]), f"SGD(lr=1) update of {name}"
            scale = avg64.abs().max().item() + 1e-30
            mine.append((res[0]["grads"][name].double() - avg64).abs().max().item() / scale)
            ref.append((avg32 - avg64).abs().max().item() / scale)
        mine, ref = np.array(mine), np.array(ref)
        assert mine.max() <= 3 * ref.max() + 2e-4 and np.median(mine) <= 3 * np.median(ref) + 1e-4, (mine.max(), ref.max(), np.median(mine), np.median(ref))

    attempts, res = [], None
    for k in range(ATTEMPTS):
        try:
            with mp.Manager() as mgr:
                out = mgr.dict()
                mp.spawn(_worker, args=(world, _free_port(), out, "gloo", batch), nprocs=world, join=True)
                res = [out[r] for r in range(world)]
            attempts.append({"ok": True})
            break
        except Exception as e:  # noqa: BLE001 -- a rank died (or raised): reported; retried once
            attempts.append({"ok": False, "error": f"{type(e).__name__}: {str(e)[:600]}"})
            if k == ATTEMPTS - 1:
                _note("eight_rank_train_step", attempts)
                raise
    _note("eight_rank_train_step", attempts)
    check(res)                                   # numbers are checked ONCE: a deviation is a failure, not a reason to run again


def test_bench_eight_rank_dress_rehearsal(gpu_device):
    """BASELINE configs[2] as the driver will launch it on an 8-GPU node -- `python bench.py --gpus 8` -- rehearsed on however many GPUs
    this box has: with fewer than eight the ranks share them over gloo (--share-gpus), so everything but RCCL itself runs: the launcher,
    the rendezvous of eight ranks, the gloo control plane, the weak leg (batch 10 per rank) and the strong leg (global batch 80 = 8
    shards of 10: the same shard), the bucketed overlap report, and the replica check after the steps."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    eight = torch.cuda.device_count() >= 8
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--mode", "train", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--strong-steps", "1"]
    if not eight:
        cmd.append("--share-gpus")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}

    def check(r):
        assert r.returncode == 0, [ln for ln in r.stderr.splitlines() if "amdgpu.ids" not in ln and "[Gloo]" not in ln][-30:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        t = json.loads(lines[0])
        assert "error" not in t, t
        assert t["n_gpus"] == 8 and t["config"]["global_batch"] == 80 and t["config"]["batch_per_gpu"] == 10 and t["config"]["parallelism"] == "dp8"
        assert t["config"]["rccl_world_size"] == 8 and t["scaling"] == "weak" and t["value"] > 0
        st = t["strong"]
        assert st["global_batch"] == 80 and st["batch_per_gpu"] == 10 and st["n_gpus"] == 8 and st["scaling"] == "strong"
        ov = t["dp_overlap"]
        assert ov is not None and len(ov["buckets"]) >= 3 and ov["allreduce_ms_total"] > 0
        assert sum(b["bytes"] for b in ov["buckets"]) >= 4 * 11_341_000
        assert t["replicas"]["identical"] is True and t["replicas"]["ranks"] == 8 and t["replicas"]["bucket_copies"] == 0, t["replicas"]
        assert t["rccl"]["ok"] and t["rccl"]["backend"] == ("nccl" if eight else "gloo")

    attempts = []
    for k in range(ATTEMPTS):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
        died = r.returncode != 0 and ("HSA_STATUS_ERROR" in r.stderr or r.returncode < 0)
        attempts.append({"ok": not died, "returncode": r.returncode, "hsa_illegal_instruction": "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" in r.stderr})
        if not died or k == ATTEMPTS - 1:
            break                                # only a rank that died (HSA abort) is run again; anything else is judged as it is
    _note("bench_eight_rank_rehearsal", attempts)
    check(r)
