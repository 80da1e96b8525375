"""Helpers of the reference's train.py with the same signatures: mixup (train.py:19-40), get_random_mask (train.py:42-57)."""
import numpy as np
import torch

from . import ops


def mixup(x, y, alpha=0.5):
    """Sample mixup: lambda ~ Beta(alpha, alpha) folded to >= 0.5 (numpy RNG) and a random partner permutation
    (torch CPU RNG) drawn on the host exactly as the reference does; the convex mix runs in one HIP kernel per tensor."""
    batch_size = x.size()[0]
    lamb = np.random.beta(alpha, alpha, size=batch_size)
    lamb = np.maximum(lamb, 1 - lamb)
    index = torch.randperm(batch_size)
    lam_d = torch.from_numpy(lamb).float().to(x.device)
    perm_d = index.to(device=x.device, dtype=torch.int32)
    return ops.mixup(x.contiguous(), lam_d, perm_d), ops.mixup(y.contiguous(), lam_d, perm_d)


def get_random_mask(mask_size, mask_ratio, device=None):
    """Bernoulli(mask_ratio) mask of shape (N, L, 1); 1 = masked (train.py:54-57)."""
    mask = np.random.binomial(1, mask_ratio, size=mask_size)
    mask = torch.from_numpy(mask).float().unsqueeze(-1)
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return mask.to(device) if device is not None else mask
