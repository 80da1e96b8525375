"""WBCELoss with the reference's signature (utils/metric.py:3-20), computed by fused HIP kernels (forward reduction
in fp64 partials, closed-form backward)."""
from ..autograd_ops import wbce_loss


def WBCELoss(y_pred, y, reduce=True):
    """Weighted binary cross entropy of TrackNetV2: mean over all elements (reduce=True) or per sample, shape (N,)."""
    return wbce_loss(y_pred, y, reduce)


def get_metric(TP, TN, FP1, FP2, FN):
    """accuracy, precision, recall, f1, miss_rate from the five prediction-type counts (utils/metric.py:22-46);
    every ratio is 0 when its denominator is 0."""
    total = TP + TN + FP1 + FP2 + FN
    accuracy = (TP + TN) / total if total > 0 else 0
    precision = TP / (TP + FP1 + FP2) if (TP + FP1 + FP2) > 0 else 0
    recall = TP / (TP + FN) if (TP + FN) > 0 else 0
    f1 = 2 * precision * recall / (precision + recall) if (precision + recall) > 0 else 0
    miss_rate = FN / (TP + FN) if (TP + FN) > 0 else 0
    return accuracy, precision, recall, f1, miss_rate
