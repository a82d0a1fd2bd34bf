"""Metrics of the reference's ``get_metrics`` (``src/models/smp/utils.py:13-36``) computed from
the integer tp/fp/fn/tn counts the Dice kernel emits (host-side ratios on [B, C] arrays), plus the
epoch aggregation quirk of ``save_metrics_on_epoch`` (``utils.py:53-73``)."""
import numpy as np


def _div(num, den, zero_division):
    num = num.astype(np.float32)
    den = den.astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        out = num / den
    return np.where(np.isnan(out), np.float32(zero_division), out).astype(np.float32)


def iou_score(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fp + fn, zero_division)


def f1_score(tp, fp, fn, tn, zero_division=1.0):
    return _div(2 * tp, 2 * tp + fn + fp, zero_division)


def precision(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fp, zero_division)


def sensitivity(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fn, zero_division)


def get_metrics_from_stats(stats, loss, eps=1e-7):
    """stats: int64 tensor [B, C, 4] = tp, fp, fn, tn (one D2H copy, as the reference's .cpu())."""
    s = stats.detach().cpu().numpy()
    tp, fp, fn, tn = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    iou = iou_score(tp, fp, fn, tn, eps)
    return {
        'loss': loss.detach().cpu().numpy(),
        'iou': iou,
        'dice': 2 * iou / (iou + 1),
        'recall': sensitivity(tp, fp, fn, tn, eps),
        'precision': precision(tp, fp, fn, tn, eps),
        'f1': f1_score(tp, fp, fn, tn, eps),
    }


def aggregate_epoch(metrics_epoch):
    """Pairwise running mean, later batches weigh more (utils.py:53-73)."""
    metrics = {}
    for name in metrics_epoch[0].keys():
        for batch in metrics_epoch:
            v = batch[name]
            if name not in metrics:
                metrics[name] = v if v.size == 1 else np.mean(v, axis=0)
            elif v.size == 1:
                metrics[name] = np.mean((np.squeeze(v), np.squeeze(metrics[name])))
            else:
                metrics[name] = np.mean((np.mean(v, axis=0), metrics[name]), axis=0)
    return metrics
