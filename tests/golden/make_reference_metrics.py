"""Golden vectors produced by the REFERENCE's own code (build container only; /root/reference never travels).

What the reference can pin of the hot path without its absent third-party packages is the epoch bookkeeping of
`src/models/smp/utils.py`: `save_metrics_on_epoch` (pairwise running mean over the batches, per-class + Mean csv rows,
best-metric tracking; utils.py:39-166) and the glue of `get_metrics` (Dice = 2 IoU / (IoU + 1), zero_division = 1e-7;
utils.py:13-36).  That module imports cv2, wandb and segmentation_models_pytorch at the top (utils.py:5-10), none of which is
installed here, so they are stubbed in `sys.modules` for the import:
  * cv2  -- an empty module (only the overlay helpers further down the file use it);
  * wandb -- `run.summary` = a dict, `log` = no-op (network logging, out of scope);
  * segmentation_models_pytorch -- `metrics.get_stats / iou_score / f1_score / precision / sensitivity` bound to the
    ORACLE's restatements (oracle/metrics.py).  The `get_metrics` vectors therefore pin the reference's glue around those five
    calls, not smp's arithmetic itself (which stays "parity unpinned").
The reference file is loaded with importlib straight from /root/reference and executed unmodified; nothing of its text is
copied.  Output: tests/golden/reference_metrics.json = the inputs (batch dicts) and what the reference returned / wrote.

    python tests/golden/make_reference_metrics.py
"""
import csv
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_UTILS = '/root/reference/src/models/smp/utils.py'


def load_reference_utils():
    sys.path.insert(0, ROOT)
    from oracle import metrics as om
    cv2 = types.ModuleType('cv2')
    wandb = types.ModuleType('wandb')
    wandb.run = types.SimpleNamespace(summary={})
    wandb.log = lambda *a, **k: None
    smp = types.ModuleType('segmentation_models_pytorch')
    smp.metrics = types.SimpleNamespace(get_stats=om.get_stats, iou_score=om.iou_score, f1_score=om.f1_score,
                                        precision=om.precision, sensitivity=om.sensitivity)
    sys.modules.update({'cv2': cv2, 'wandb': wandb, 'segmentation_models_pytorch': smp})
    spec = importlib.util.spec_from_file_location('reference_smp_utils', REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, wandb


def batch_dict(rng, n_img, n_cls):
    """A dict shaped like get_metrics' output: loss 0-d, the rest [n_img, n_cls] float32."""
    iou = rng.random((n_img, n_cls)).astype(np.float32)
    return {'loss': np.array(rng.random(), dtype=np.float32), 'iou': iou, 'dice': (2 * iou / (iou + 1)).astype(np.float32),
            'recall': rng.random((n_img, n_cls)).astype(np.float32), 'precision': rng.random((n_img, n_cls)).astype(np.float32),
            'f1': rng.random((n_img, n_cls)).astype(np.float32)}


def to_json(d):
    return {k: np.asarray(v).tolist() for k, v in d.items()}


def main():
    ref, wandb = load_reference_utils()
    out = {'source': 'src/models/smp/utils.py (save_metrics_on_epoch :39-166, get_metrics :13-36), executed in the build container',
           'epoch_cases': [], 'get_metrics_cases': []}
    rng = np.random.default_rng(20240607)
    cwd = os.getcwd()
    # ---- save_metrics_on_epoch: 1-class and 2-class models, several epochs each, ragged last batch
    for name, classes, batches_per_epoch in (('one_class', ['Lumen'], [3, 1, 4]), ('two_class', ['Lipid core', 'Fibrous cap'], [2, 5, 3])):
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            os.makedirs(f'models/{name}')
            wandb.run.summary.clear()
            best = {}
            epochs = []
            for epoch, nb in enumerate(batches_per_epoch, start=1):
                sizes = [4] * (nb - 1) + [3]                       # last batch of an epoch is smaller
                batches = [batch_dict(rng, s, len(classes)) for s in sizes]
                for split in ('test', 'train'):                     # the reference logs 'test' (validation) before 'train'
                    ret = ref.save_metrics_on_epoch(metrics_epoch=batches, split=split, model_name=name, classes=classes, epoch=epoch,
                                                    best_metrics=best if split == 'test' else None)
                    if split == 'test':
                        best = ret
                epochs.append({'epoch': epoch, 'batches': [to_json(b) for b in batches],
                               'best_after': {k: {'value': float(v['value']), 'epoch': int(v['epoch'])} for k, v in best.items()}})
            with open(f'models/{name}/metrics.csv', newline='') as f:
                rows = list(csv.DictReader(f))
            os.chdir(cwd)
        out['epoch_cases'].append({'name': name, 'classes': classes, 'epochs': epochs, 'csv_rows': rows,
                                   'wandb_summary': {k: float(v) for k, v in wandb.run.summary.items()}})
    # ---- get_metrics glue (smp.metrics stubbed with the oracle's restatement, see the module docstring)
    g = torch.Generator().manual_seed(5)
    for n_img, n_cls, hw in ((3, 1, 16), (2, 2, 12), (2, 4, 8)):
        pred = (torch.rand(n_img, n_cls, hw, hw, generator=g) > 0.5).float()
        mask = (torch.rand(n_img, n_cls, hw, hw, generator=g) > 0.6).float()
        mask[0, -1] = 0; pred[0, -1] = 0                            # an empty class: 0/0 -> zero_division = 1e-7
        loss = torch.tensor(0.25 + 0.1 * n_cls)
        m = ref.get_metrics(mask=mask, pred_mask=pred, loss=loss)
        out['get_metrics_cases'].append({'pred': pred.numpy().astype(np.uint8).tolist(), 'mask': mask.numpy().astype(np.uint8).tolist(),
                                         'loss': float(loss), 'expected': to_json(m)})
    with open(os.path.join(HERE, 'reference_metrics.json'), 'w') as f:
        json.dump(out, f)
    print('wrote reference_metrics.json:', [(c['name'], len(c['csv_rows'])) for c in out['epoch_cases']], len(out['get_metrics_cases']))


if __name__ == '__main__':
    main()
