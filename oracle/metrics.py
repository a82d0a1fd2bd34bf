"""Oracle metrics: smp 0.3.3 ``metrics.functional`` (multilabel) restated plus
the reference's ``get_metrics`` (``src/models/smp/utils.py:13-36``).
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch


def get_stats(output, target, mode='multilabel'):
    """tp/fp/fn/tn per (image, class), int64 -- smp ``_get_stats_multilabel``."""
    assert mode == 'multilabel'
    assert output.shape == target.shape
    b, c = output.shape[:2]
    output = output.reshape(b, c, -1).long()
    target = target.reshape(b, c, -1).long()
    tp = (output * target).sum(2)
    fp = output.sum(2) - tp
    fn = target.sum(2) - tp
    tn = output.shape[2] - (tp + fp + fn)
    return tp, fp, fn, tn


def _div(num, den, zero_division):
    num = num.to(torch.float32)
    den = den.to(torch.float32)
    out = num / den
    return torch.where(torch.isnan(out), torch.tensor(float(zero_division)), out)


def iou_score(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fp + fn, zero_division)


def f1_score(tp, fp, fn, tn, zero_division=1.0):
    # f-beta with beta=1: (1+b2) tp / ((1+b2) tp + b2 fn + fp)
    return _div(2 * tp, 2 * tp + fn + fp, zero_division)


def precision(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fp, zero_division)


def sensitivity(tp, fp, fn, tn, zero_division=1.0):
    return _div(tp, tp + fn, zero_division)


def get_metrics(mask, pred_mask, loss, eps=1e-7):
    """Reference ``src/models/smp/utils.py:13-36`` restated."""
    tp, fp, fn, tn = get_stats(pred_mask.long(), mask.long(), mode='multilabel')
    iou = iou_score(tp, fp, fn, tn, zero_division=eps)
    dice = 2 * iou.numpy() / (iou.numpy() + 1)
    return {
        'loss': loss.detach().cpu().numpy(),
        'iou': iou.numpy(),
        'dice': dice,
        'recall': sensitivity(tp, fp, fn, tn, zero_division=eps).numpy(),
        'precision': precision(tp, fp, fn, tn, zero_division=eps).numpy(),
        'f1': f1_score(tp, fp, fn, tn, zero_division=eps).numpy(),
    }


def aggregate_epoch(metrics_epoch):
    """The pairwise running mean of ``save_metrics_on_epoch``
    (reference ``src/models/smp/utils.py:53-73``): later batches weigh more."""
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
