"""CPU: tests/oracle_pool.py (the worker-process pool the full-size GPU parity tests evaluate their host oracles in) returns what the
in-process oracle returns -- the spawn path included (forced here: this box has too few cores for the pool to engage by itself)."""
import os

import torch

import oracle_pool
from oracle import nets


def test_pool_results_equal_the_in_process_oracle(monkeypatch):
    spec = dict(in_dim=9, out_dim=3, seed=31, gain=2.4, var_range=None, n=1, h=16, w=32)
    specs = [dict(spec, kind="train", dtype="float64"), dict(spec, kind="train", dtype="float32"),
             dict(spec, kind="eval", dtype="float64", var_range=(0.5, 2.0))]
    here = oracle_pool.run([dict(s) for s in specs], workers=1)
    monkeypatch.setattr(os, "cpu_count", lambda: 64)          # lets run() engage two spawned workers
    pooled = oracle_pool.run([dict(s) for s in specs], workers=2)
    # (not bit-wise: a worker runs with another thread count than this process, and ATen's reductions follow it)
    for (a, b), tol in zip(zip(here[:2], pooled[:2]), (1e-11, 1e-4)):
        assert torch.allclose(a[0], b[0], rtol=tol, atol=0) and torch.allclose(a[1], b[1], rtol=tol, atol=tol * 1e-2)
        assert list(a[2]) == list(b[2])
        for k in a[2]:
            assert (a[2][k] - b[2][k]).abs().max() <= tol * a[2][k].abs().max() + 1e-300, k
    assert torch.allclose(here[2], pooled[2], rtol=1e-11, atol=1e-12)
    # and what a spec means: the oracle's own functions on the oracle's own synthetic state
    sd = nets.synth_state(nets.tracknet_state_shapes(9, 3), 31, calibrated=True)
    x, y = nets.synth_input((1, 9, 16, 32), 1031), nets.disc_heatmaps(1, 3, 16, 32, 2031)
    l, p, g, _ = nets.tracknet_train_step_grads(sd, x, y, torch.float64)
    torch.set_num_threads(int(oracle_pool._threads(1)))
    l, p, g, _ = nets.tracknet_train_step_grads(sd, x, y, torch.float64)
    assert torch.allclose(l, here[0][0], rtol=1e-11) and torch.allclose(p, here[0][1], rtol=1e-11, atol=1e-13)
    assert all((g[k] - here[0][2][k]).abs().max() <= 1e-11 * g[k].abs().max() for k in g)
