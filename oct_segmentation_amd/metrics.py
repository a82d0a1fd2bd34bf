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


def _metrics_from_numpy(s, loss, eps=1e-7):
    tp, fp, fn, tn = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
    iou = iou_score(tp, fp, fn, tn, eps)
    return {
        'loss': loss,
        'iou': iou,
        'dice': 2 * iou / (iou + 1),
        'recall': sensitivity(tp, fp, fn, tn, eps),
        'precision': precision(tp, fp, fn, tn, eps),
        'f1': f1_score(tp, fp, fn, tn, eps),
    }


def get_metrics_from_stats(stats, loss, eps=1e-7):
    """stats: int64 tensor [B, C, 4] = tp, fp, fn, tn (one D2H copy, as the reference's .cpu())."""
    return _metrics_from_numpy(stats.detach().cpu().numpy(), loss.detach().cpu().numpy(), eps)


class DeferredMetrics:
    """On-device accumulation of a split's step records: the reference's ``get_metrics`` copies tp / fp / fn / tn
    and the loss to the host in EVERY step (``src/models/smp/utils.py:25-35`` -- a device sync per step).  Here the int64 counts
    [B, C, 4] and the f32 loss of each step stay in HBM; ``flush()`` stacks them, crosses to the host once and builds the very
    per-step dicts ``get_metrics_from_stats`` would have built (same integer counts, same float32 ratios), so the epoch rows of
    ``save_metrics_on_epoch`` -- pairwise running mean included -- are bit-identical to the per-step path."""

    def __init__(self):
        self._stats, self._loss = [], []

    def append(self, stats, loss):
        self._stats.append(stats.detach())          # fresh tensors of the Dice launch: nothing overwrites them
        self._loss.append(loss.detach().reshape(1))

    def __len__(self):
        return len(self._stats)

    def flush(self, eps=1e-7):
        import torch
        if not self._stats:
            return []
        shapes = [tuple(s.shape) for s in self._stats]
        flat = torch.cat([s.reshape(-1) for s in self._stats]).cpu().numpy()      # ONE copy of every count of the epoch
        losses = torch.cat(self._loss).cpu().numpy()                                # ... and one of the losses
        out, off = [], 0
        for i, shp in enumerate(shapes):
            n = int(np.prod(shp))
            out.append(_metrics_from_numpy(flat[off:off + n].reshape(shp), losses[i].reshape(()), eps))
            off += n
        self._stats, self._loss = [], []
        return out


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


CSV_FIELDS = ['Epoch', 'Loss', 'IoU', 'Dice', 'Precision', 'Recall', 'F1', 'Split', 'Class']


def save_metrics_on_epoch(metrics_epoch, split, model_dir, classes, epoch, best_metrics=None):
    """Epoch bookkeeping of the reference without W&B (``src/models/smp/utils.py:39-166``): aggregate the epoch's batch
    dicts with the pairwise running mean, append one row per class plus a ``Mean`` row to ``<model_dir>/metrics.csv``
    (header on first use, same column names), keep the best IoU / Dice / Precision / Recall with their epochs.
    Values stay what the reference's arithmetic makes them (numpy float32 scalars written with ``str()``), so the file equals
    the reference's byte for byte on the same batch dicts (tests/golden/reference_metrics.json).
    Returns ``(summary_dict, best_metrics)``; the summary carries the ``<split>/<metric>`` keys the reference logs to W&B."""
    import csv
    import os
    m = aggregate_epoch(metrics_epoch)
    names = ('iou', 'dice', 'precision', 'recall', 'f1')
    summary = {f'{split}/loss': m['loss']}
    for k in ('iou', 'dice', 'precision', 'recall', 'f1'):
        summary[f'{split}/{k}'] = m[k].mean()
    if best_metrics is not None:
        for k in ('iou', 'dice', 'precision', 'recall'):
            cur = summary[f'{split}/{k}']
            if k not in best_metrics or cur > best_metrics[k]['value']:
                best_metrics[k] = {'value': cur, 'epoch': epoch}
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, 'metrics.csv')
    new_file = not os.path.exists(path)
    multi = len(classes) > 1
    with open(path, 'a', newline='') as f:
        w = csv.DictWriter(f, fieldnames=CSV_FIELDS)
        if new_file:
            w.writeheader()
        for i, cl in enumerate(classes):
            pick = (lambda v: v[i] if multi else v)      # one class: the whole (length-1) array, as the reference writes it
            for k in names:
                summary[f'{split}/{k} ({cl})'] = pick(m[k])
            w.writerow({'Epoch': epoch, 'Loss': m['loss'], 'IoU': pick(m['iou']), 'Dice': pick(m['dice']), 'Precision': pick(m['precision']),
                        'Recall': pick(m['recall']), 'F1': pick(m['f1']), 'Split': split, 'Class': cl})
        w.writerow({'Epoch': epoch, 'Loss': m['loss'], 'IoU': m['iou'].mean(), 'Dice': m['dice'].mean(), 'Precision': m['precision'].mean(),
                    'Recall': m['recall'].mean(), 'F1': m['f1'].mean(), 'Split': split, 'Class': 'Mean'})
    return summary, best_metrics
