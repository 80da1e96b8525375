"""WBCELoss with the reference's signature (utils/metric.py:3-20), computed by fused HIP kernels (forward reduction
in fp64 partials, closed-form backward)."""
from ..autograd_ops import wbce_loss


def WBCELoss(y_pred, y, reduce=True):
    """Weighted binary cross entropy of TrackNetV2: mean over all elements (reduce=True) or per sample, shape (N,)."""
    return wbce_loss(y_pred, y, reduce)
