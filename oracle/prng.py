"""Portable counter-based PRNG (splitmix64) used for every synthetic tensor.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The generator depends only on
numpy integer arithmetic, so the golden-vector script (run in the survey
container, where the reference is importable) and the GPU box regenerate
bit-identical weights and inputs without depending on torch's generator.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed, lo=0.0, hi=1.0):
    """float32 array ~ U[lo, hi): element i is a pure function of (seed, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(1))
        bits = _splitmix64(idx ^ key)
    u = (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))  # 24-bit mantissa
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def name_seed(name, seed):
    """Stable per-tensor seed from a state_dict key."""
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF
