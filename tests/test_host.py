"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the graph
builder reproduces the oracle's parameter tree / MAC counts, argument validation mirrors the reference
errors, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from oct_segmentation_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    lib = L.lib()
    header = open(os.path.join(ROOT, 'include', 'octseg.h')).read()
    declared = set(re.findall(r'\b(octseg_[a-z0-9_]+)\s*\(', header))
    declared -= {'octseg_plan'}  # the opaque struct tag
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/octseg.h but not exported'
        assert name in L.SYMBOLS, f'{name} has no ctypes signature in _lib.SYMBOLS'
    assert lib.octseg_version() >= 100


def _plan(arch, enc, classes, B, H, W, dt=L.BF16):
    d = L.NetDesc(arch.encode(), enc.encode(), classes, B, H, W, dt)
    p = C.c_void_p()
    rc = L.lib().octseg_plan_create(C.byref(d), C.byref(p))
    return rc, p


@pytest.mark.parametrize('arch,enc,classes,S,gmac', [
    ('unet', 'resnet18', 1, 256, 5.40), ('unetplusplus', 'resnet101', 1, 704, 471.07), ('linknet', 'resnet50', 2, 704, 54.99),
    ('unet', 'resnet50', 1, 704, 80.41), ('unetplusplus', 'resnet34', 1, 704, 139.03), ('linknet', 'resnet18', 2, 704, 19.78)])
def test_graph_matches_oracle_tree_and_survey_macs(arch, enc, classes, S, gmac):
    from oracle import create_model
    lib = L.lib()
    rc, p = _plan(arch, enc, classes, 1, S, S)
    assert rc == 0, lib.octseg_last_error()
    try:
        names = {}
        for i in range(lib.octseg_plan_num_params(p)):
            pi = L.ParamInfo()
            assert lib.octseg_plan_param_info(p, i, C.byref(pi)) == 0
            names[pi.name.decode()] = pi
        sd = create_model(arch, enc, classes=classes).state_dict()
        params = {k: v for k, v in sd.items() if 'running_' not in k and 'num_batches' not in k}
        assert set(params) == set(names)
        for k, v in params.items():
            pi = names[k]
            if pi.kind == L.P_STEM:
                assert v.numel() == pi.O * 147 and pi.numel == pi.O * pi.KP
            else:
                assert v.numel() == pi.numel, k
            if pi.kind == L.P_CONV:
                assert tuple(v.shape) == (pi.O, pi.I, pi.R, pi.S)
            if pi.kind == L.P_CONVT:
                assert tuple(v.shape) == (pi.I, pi.O, pi.R, pi.S)
        bns = set()
        for i in range(lib.octseg_plan_num_bn(p)):
            bi = L.BNInfo()
            assert lib.octseg_plan_bn_info(p, i, C.byref(bi)) == 0
            bns.add(bi.name.decode())
        assert bns == {k[:-len('.running_mean')] for k in sd if k.endswith('.running_mean')}
        assert lib.octseg_plan_fwd_macs(p) / 1e9 == pytest.approx(gmac, abs=0.02)   # SURVEY.md Appendix B / BASELINE.md
        assert lib.octseg_plan_workspace_bytes(p) > 0
    finally:
        lib.octseg_plan_destroy(p)


def test_plan_argument_validation():
    lib = L.lib()
    rc, _ = _plan('unet', 'resnet18', 1, 1, 100, 100)
    assert rc == -1 and b'divisible by 32' in lib.octseg_last_error()   # smp check_input_shape text
    rc, _ = _plan('fpn', 'resnet18', 1, 1, 64, 64)
    assert rc == -3
    rc, _ = _plan('unet', 'vgg16', 1, 1, 64, 64)
    assert rc == -3
    rc, _ = _plan('unet', 'resnet18', 1, 1, 64, 64, dt=7)
    assert rc == -2
    assert lib.octseg_optim_step(9, None, None, None, None, 0, 0.0, 0.0, 1, 1.0, None) == -5


def test_host_api_errors_mirror_the_reference():
    from oct_segmentation_amd.engine import get_preprocessing_params
    assert get_preprocessing_params('resnet50')['mean'] == [0.485, 0.456, 0.406]
    with pytest.raises(KeyError):
        get_preprocessing_params('vgg16')


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_product_path_has_no_cpu_fallback():
    from oct_segmentation_amd.engine import SegNet
    with pytest.raises((RuntimeError, AssertionError)):
        net = SegNet('unet', 'resnet18', classes=1, device='cuda')  # torch raises: no GPU to allocate the arena on
        net(torch.zeros(1, 3, 64, 64))


def test_metrics_from_stats_match_oracle():
    from oct_segmentation_amd.metrics import get_metrics_from_stats
    from oracle import get_metrics, get_stats
    g = torch.Generator().manual_seed(0)
    pred = (torch.rand(3, 2, 16, 16, generator=g) > 0.5).float()
    mask = (torch.rand(3, 2, 16, 16, generator=g) > 0.7).float()
    mask[1, 1] = 0
    pred[1, 1] = 0   # 0/0 -> zero_division = 1e-7
    tp, fp, fn, tn = get_stats(pred.long(), mask.long())
    ours = get_metrics_from_stats(torch.stack([tp, fp, fn, tn], dim=-1), torch.tensor(0.25))
    ref = get_metrics(mask, pred, torch.tensor(0.25))
    for k in ref:
        assert ours[k] == pytest.approx(ref[k], rel=1e-6, abs=1e-9), k


def test_pil_nearest_index_table_is_pillows_rule():
    """The index tables handed to octseg_mask_assemble reproduce PIL.Image.resize(..., NEAREST) exactly, including the
    positions that fall on an integer (Pillow accumulates the step in double and truncates)."""
    import numpy as np
    from PIL import Image
    from oct_segmentation_amd.predict import pil_nearest_index
    for src in (64, 224, 512, 704, 100, 37):
        img = Image.fromarray(np.tile(np.arange(src, dtype=np.float32), (2, 1)))
        for dst in (96, 48, 37, 100, 704, 512, 300, 33, 63, 65, 1000):
            want = np.asarray(img.resize((dst, 2), Image.NEAREST))[0].astype(np.int64)
            assert np.array_equal(pil_nearest_index(src, dst), want), (src, dst)


def test_metrics_csv_bookkeeping(tmp_path):
    """metrics.csv schema, per-class + Mean rows, pairwise-mean aggregation and best-metric tracking (utils.py:39-158)."""
    import csv
    import numpy as np
    from oct_segmentation_amd.metrics import save_metrics_on_epoch, CSV_FIELDS
    classes = ['Lipid core', 'Fibrous cap']
    def batch(loss, iou):
        iou = np.asarray(iou, dtype=np.float64)
        return {'loss': np.array(loss), 'iou': iou, 'dice': 2 * iou / (iou + 1), 'recall': iou, 'precision': iou, 'f1': iou}
    b = [batch(1.0, [[0.2, 0.4], [0.4, 0.8]]), batch(0.0, [[0.8, 0.0], [0.8, 0.0]]), batch(0.5, [[0.0, 0.0], [0.0, 0.0]])]
    best = {}
    s1, best = save_metrics_on_epoch(b, 'test', str(tmp_path), classes, 1, best)
    assert s1['test/loss'] == pytest.approx(((1.0 + 0.0) / 2 + 0.5) / 2)                 # later batches weigh more
    assert s1['test/iou (Lipid core)'] == pytest.approx(((0.3 + 0.8) / 2 + 0.0) / 2)
    assert s1['test/iou'] == pytest.approx((s1['test/iou (Lipid core)'] + s1['test/iou (Fibrous cap)']) / 2)
    s2, best = save_metrics_on_epoch(b[:2], 'test', str(tmp_path), classes, 2, best)
    assert best['iou'] == {'value': s2['test/iou'], 'epoch': 2} and s2['test/iou'] > s1['test/iou']
    rows = list(csv.DictReader(open(tmp_path / 'metrics.csv')))
    assert list(rows[0].keys()) == CSV_FIELDS and len(rows) == 6
    assert [r['Class'] for r in rows[:3]] == ['Lipid core', 'Fibrous cap', 'Mean'] and rows[3]['Epoch'] == '2' and rows[0]['Split'] == 'test'
    assert float(rows[2]['IoU']) == pytest.approx(s1['test/iou'])
