"""What the reference itself can pin of this path (CPU tests; fixtures generated in the build container by
tests/golden/make_reference_metrics.py, which EXECUTES the reference's src/models/smp/utils.py, plus an excerpt of a
result file the reference publishes):

  * save_metrics_on_epoch (utils.py:39-166): pairwise running mean, per-class + Mean rows, best-metric tracking --
    the product's metrics.csv must equal the reference's byte for byte on the same batch dicts, and the oracle's
    aggregate_epoch must reproduce the aggregated numbers;
  * get_metrics (utils.py:13-36): Dice = 2 IoU / (IoU + 1), zero_division = 1e-7 (glue pinned; the five smp.metrics calls inside
    it were bound to the oracle's restatement when the fixture was made -- smp itself is absent -- so those stay unpinned);
  * eval/training/Lumen/fold_1/metrics.csv (published training log): per row Dice == F1 to float32 rounding, i.e. the per-image
    Dice the reference derives from IoU IS the F1 score 2tp / (2tp + fp + fn); the oracle and the product must have that identity.
"""
import csv
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def fx():
    with open(os.path.join(GOLDEN, 'reference_metrics.json')) as f:
        return json.load(f)


def _batches(ep):
    return [{k: np.asarray(v, dtype=np.float32) for k, v in b.items()} for b in ep['batches']]


def test_metrics_csv_equals_the_reference_byte_for_byte(fx, tmp_path):
    from oct_segmentation_amd.metrics import save_metrics_on_epoch
    for case in fx['epoch_cases']:
        d = tmp_path / case['name']
        best = {}
        for ep in case['epochs']:
            for split in ('test', 'train'):     # model.py:134-148 then :97-106 -- validation ('test') rows come first
                summary, b = save_metrics_on_epoch(_batches(ep), split, str(d), case['classes'], ep['epoch'], best if split == 'test' else None)
                if split == 'test':
                    best = b
            assert {k: {'value': float(v['value']), 'epoch': int(v['epoch'])} for k, v in best.items()} == ep['best_after']
        with open(d / 'metrics.csv', newline='') as f:
            rows = list(csv.DictReader(f))
        assert rows == case['csv_rows'], case['name']
        # what the reference leaves in wandb.run.summary = the best values and their epochs
        for k in ('iou', 'dice', 'precision', 'recall'):
            assert float(best[k]['value']) == case['wandb_summary'][f'best_{k}'] and best[k]['epoch'] == case['wandb_summary'][f'best_{k}_epoch']


def test_oracle_aggregate_epoch_reproduces_the_reference_rows(fx):
    from oracle.metrics import aggregate_epoch
    for case in fx['epoch_cases']:
        rows = [r for r in case['csv_rows'] if r['Split'] == 'test']
        per_epoch = len(case['classes']) + 1
        for i, ep in enumerate(case['epochs']):
            m = aggregate_epoch(_batches(ep))
            mean_row = rows[i * per_epoch + per_epoch - 1]
            assert mean_row['Class'] == 'Mean' and int(mean_row['Epoch']) == ep['epoch']
            assert str(m['loss']) == mean_row['Loss']
            for col, key in (('IoU', 'iou'), ('Dice', 'dice'), ('Precision', 'precision'), ('Recall', 'recall'), ('F1', 'f1')):
                assert str(m[key].mean()) == mean_row[col], (case['name'], ep['epoch'], col)
                for c in range(len(case['classes'])):
                    want = rows[i * per_epoch + c][col]
                    got = m[key][c] if len(case['classes']) > 1 else m[key]
                    assert str(got) == want


def test_get_metrics_glue_matches_the_reference(fx):
    from oracle import get_metrics, get_stats
    from oct_segmentation_amd.metrics import get_metrics_from_stats
    for case in fx['get_metrics_cases']:
        pred = torch.tensor(case['pred'], dtype=torch.float32)
        mask = torch.tensor(case['mask'], dtype=torch.float32)
        loss = torch.tensor(case['loss'])
        want = {k: np.asarray(v, dtype=np.float32) for k, v in case['expected'].items()}
        ours_oracle = get_metrics(mask, pred, loss)
        tp, fp, fn, tn = get_stats(pred.long(), mask.long())
        ours_product = get_metrics_from_stats(torch.stack([tp, fp, fn, tn], dim=-1), loss)
        for k, w in want.items():
            assert np.array_equal(np.asarray(ours_oracle[k], dtype=np.float32), w), k
            assert np.array_equal(np.asarray(ours_product[k], dtype=np.float32), w), k
        assert (want['iou'] == np.float32(1e-7)).any()          # the empty class went through zero_division


def test_published_training_log_identity_dice_equals_f1():
    """The reference's own published rows: Dice (derived from IoU per image, utils.py:25) equals F1 (smp f1_score) to float32
    rounding in every row -- and so must the oracle's and the product's per-image metrics."""
    with open(os.path.join(GOLDEN, 'lumen_fold1_metrics_excerpt.csv'), newline='') as f:
        rows = list(csv.DictReader(f))
    assert len(rows) == 20 and rows[0]['Class'] == 'Lumen' and {r['Split'] for r in rows} == {'test', 'train'}
    for r in rows:
        d, f1, iou = float(r['Dice']), float(r['F1']), float(r['IoU'])
        assert abs(d - f1) <= 3e-7 * max(d, f1), r
        assert d <= 2 * iou / (iou + 1) + 1e-6      # 2x / (1 + x) is concave: the mean of per-image Dice is <= Dice of the mean IoU (Jensen)
        assert 0.0 < float(r['Loss']) < 1.0
    assert rows[0]['Split'] == 'test' and rows[1]['Split'] == 'train'     # validation rows precede training rows
    from oracle import get_metrics
    from oct_segmentation_amd.metrics import get_metrics_from_stats
    from oracle import get_stats
    g = torch.Generator().manual_seed(1)
    pred = (torch.rand(5, 2, 32, 32, generator=g) > 0.4).float()
    mask = (torch.rand(5, 2, 32, 32, generator=g) > 0.5).float()
    m = get_metrics(mask, pred, torch.tensor(0.1))
    assert np.allclose(m['dice'], m['f1'], rtol=3e-7, atol=0)
    tp, fp, fn, tn = get_stats(pred.long(), mask.long())
    p = get_metrics_from_stats(torch.stack([tp, fp, fn, tn], dim=-1), torch.tensor(0.1))
    assert np.allclose(p['dice'], p['f1'], rtol=3e-7, atol=0)


def test_epoch_rows_are_written_in_the_reference_order(fx, tmp_path):
    """The epoch bookkeeping of the train loop (train.write_epoch_rows, what fit() calls) against the reference's file: Lightning runs
    on_validation_epoch_end (model.py:134-148, 'test' rows + best metrics) BEFORE on_train_epoch_end (model.py:97-106, 'train' rows);
    a whole csv must equal the reference's byte for byte, row order included."""
    from oct_segmentation_amd.train import write_epoch_rows
    for case in fx['epoch_cases']:
        d = tmp_path / ('order_' + case['name'])
        best = {}
        for ep in case['epochs']:
            best = write_epoch_rows(str(d), case['classes'], ep['epoch'], _batches(ep), _batches(ep), best)
        with open(d / 'metrics.csv', newline='') as f:
            rows = list(csv.DictReader(f))
        assert rows == case['csv_rows'], case['name']
        per = len(case['classes']) + 1
        assert [r['Split'] for r in rows[:2 * per]] == ['test'] * per + ['train'] * per


def test_deferred_metrics_equal_the_per_step_dicts(fx, tmp_path):
    """One host copy per epoch (metrics.DeferredMetrics) must give the dicts the per-step path gives: same counts, same float32
    ratios, hence the same csv bytes (reference: the per-step .cpu() of utils.py:25-35 is what is being removed)."""
    from oct_segmentation_amd.metrics import DeferredMetrics, get_metrics_from_stats, save_metrics_on_epoch
    g = torch.Generator().manual_seed(11)
    steps = []
    for i in range(5):
        B = 1 + i % 3
        tp = torch.randint(0, 50, (B, 2), generator=g)
        fp = torch.randint(0, 50, (B, 2), generator=g)
        fn = torch.randint(0, 50, (B, 2), generator=g)
        if i == 2:
            tp[0, 1] = fp[0, 1] = fn[0, 1] = 0          # an empty class: zero_division path
        tn = 4096 - tp - fp - fn
        steps.append((torch.stack([tp, fp, fn, tn], dim=-1), torch.rand((), generator=g)))
    acc = DeferredMetrics()
    eager = []
    for s, l in steps:
        acc.append(s, l)
        eager.append(get_metrics_from_stats(s, l))
    assert len(acc) == 5
    late = acc.flush()
    assert len(acc) == 0 and len(late) == len(eager)
    for a, b in zip(late, eager):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]) and np.asarray(a[k]).dtype == np.asarray(b[k]).dtype and np.shape(a[k]) == np.shape(b[k]), k
    save_metrics_on_epoch(late, 'test', str(tmp_path / 'a'), ['x', 'y'], 1, {})
    save_metrics_on_epoch(eager, 'test', str(tmp_path / 'b'), ['x', 'y'], 1, {})
    assert (tmp_path / 'a' / 'metrics.csv').read_bytes() == (tmp_path / 'b' / 'metrics.csv').read_bytes()
