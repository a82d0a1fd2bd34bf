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
    ('unet', 'resnet50', 1, 704, 80.41), ('unetplusplus', 'resnet34', 1, 704, 139.03), ('linknet', 'resnet18', 2, 704, 19.78),
    ('unet', 'resnet152', 1, 704, 153.74)])   # (resnet152: torchvision 11.51 GMAC @224^2 -> x9.88 at 704^2, + the U-Net/r50 decoder)
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


def test_executed_macs_of_the_tied_decoder_layers(monkeypatch):
    """octseg_plan_exec_macs: the reference graph's count never moves; the tied passes of the nearest-x2 + 3x3 decoder layers
    execute 4/9 of it over the upsampled source's channels.  U-Net++/resnet101 at 704^2: 182.7 of 471.07 GMAC lie on upsampled channels
    (DESIGN.md section 7.7), so a tied pass executes 471.07 - 182.7 * 5 / 9 less what the narrow layers (below 64 channels) keep."""
    lib = L.lib()

    def macs(env, dt=L.BF16, arch='unetplusplus', enc='resnet101'):
        if env is None:
            monkeypatch.delenv('OCTSEG_TIED', raising=False)
        else:
            monkeypatch.setenv('OCTSEG_TIED', env)
        rc, p = _plan(arch, enc, 1, 1, 704, 704, dt)
        assert rc == 0, lib.octseg_last_error()
        try:
            out = (C.c_double * 3)()
            assert lib.octseg_plan_exec_macs(p, out) == 0
            return lib.octseg_plan_fwd_macs(p), tuple(out)
        finally:
            lib.octseg_plan_destroy(p)
    alg, ex = macs(None)                                   # default: the two gradients
    assert alg / 1e9 == pytest.approx(471.07, abs=0.02)
    assert ex[0] == alg and ex[1] == ex[2] and 0.77 * alg < ex[1] < 0.80 * alg     # 370.86 GMAC: -21.3 % of every conv MAC of the pass
    assert (alg - ex[1]) / 1e9 == pytest.approx(182.7 * 5 / 9, rel=0.02)
    assert macs('0') == (alg, (alg, alg, alg))
    assert macs('f')[1] == (ex[1], alg, alg)
    assert macs('fdw')[1] == (ex[1], ex[1], ex[1])
    assert macs(None, L.F32)[1] == (alg, alg, alg)          # fp32 plans keep the reference's summation
    a2, e2 = macs(None, arch='linknet', enc='resnet50')     # no nearest-x2 + 3x3 layer in the graph
    assert e2 == (a2, a2, a2)
    assert lib.octseg_plan_exec_macs(None, (C.c_double * 3)()) != 0


def test_plan_argument_validation():
    lib = L.lib()
    rc, _ = _plan('unet', 'resnet18', 1, 1, 100, 100)
    assert rc == -1 and b'divisible by 32' in lib.octseg_last_error()   # smp check_input_shape text
    rc, _ = _plan('segformer', 'resnet18', 1, 1, 64, 64)   # (not one of smp 0.3.3's nine architectures)
    assert rc == -3
    for enc in ('timm-regnetx_002', 'efficientnet-b0'):    # pairs the builder refuses: PAN dilates its encoder, LinkNet quarters the feature widths
        for arch in ('pan', 'linknet', 'deeplabv3plus'):
            rc, _ = _plan(arch, enc, 1, 1, 64, 64)
            assert rc == -3, (arch, enc)
    for arch in ('fpn', 'deeplabv3plus', 'DeepLabV3Plus', 'PSPNet', 'DeepLabV3', 'MAnet', 'PAN'):  # sweep architectures (case-insensitive like smp): round 3, MAnet / PAN round 4
        rc, pf = _plan(arch, 'resnet18', 1, 1, 64, 64)
        assert rc == 0
        lib.octseg_plan_destroy(pf)
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


def test_cv2_nearest_index_is_opencv_resizeNN():
    """segment()'s mask resize is cv2.resize(..., INTER_NEAREST) (reference src/predict.py:92-96): the index tables handed to
    octseg_mask_assemble equal the oracle's scalar restatement of OpenCV's resizeNN bit for bit -- including the sizes the
    reference runs (704 -> 1000, 512 -> 1000) and the dataset's 704 -> 750 -- and differ from Pillow's centre rule."""
    import numpy as np
    from oracle.cv2_resize import resize_nn
    from oct_segmentation_amd.predict import cv2_nearest_index, pil_nearest_index
    for src, dst in ((704, 1000), (512, 1000), (704, 750), (64, 96), (64, 48), (64, 37), (100, 33), (37, 100), (896, 1000), (7, 7)):
        ramp = np.arange(src, dtype=np.float32)
        want_cols = resize_nn(np.tile(ramp, (2, 1)), (dst, 2))[0].astype(np.int64)          # dsize = (width, height)
        want_rows = resize_nn(np.tile(ramp[:, None], (1, 2)), (2, dst))[:, 0].astype(np.int64)
        got = cv2_nearest_index(src, dst)
        assert got.dtype == np.int32 and np.array_equal(got, want_cols) and np.array_equal(got, want_rows), (src, dst)
        assert got[0] == 0 and got.max() <= src - 1
    assert (cv2_nearest_index(704, 1000) != pil_nearest_index(704, 1000)).sum() > 300   # the two rules are not interchangeable
    # a 2-D mask through the tables == the oracle's resizeNN of the mask
    rng = np.random.default_rng(3)
    m = (rng.random((64, 48, 2)) > 0.5).astype(np.float32)
    assert np.array_equal(m[cv2_nearest_index(64, 100)][:, cv2_nearest_index(48, 75)], resize_nn(m, (75, 100)))


def test_cv2_linear_u8_matches_oracle_loops():
    """preprocessing_img's cv2.resize(image_u8, (S, S)) (reference src/data/utils.py:159-166): the vectorised numpy tables against the
    oracle's scalar restatement of OpenCV's fixed-point bilinear, up- and down-scaling, odd sizes; identity is exact."""
    import numpy as np
    from PIL import Image
    from oracle.cv2_resize import resize_linear_u8
    from oct_segmentation_amd.predict import cv2_resize_linear_u8, preprocessing_img
    rng = np.random.default_rng(0)
    for (h, w, dh, dw) in ((20, 30, 17, 23), (20, 30, 41, 37), (75, 75, 64, 64), (13, 7, 26, 14), (50, 50, 32, 32)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(cv2_resize_linear_u8(a, dw, dh), resize_linear_u8(a, (dw, dh))), (h, w, dh, dw)
    a = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert np.array_equal(cv2_resize_linear_u8(a, 32, 32), a)
    out = preprocessing_img(Image.fromarray(a), 32)
    assert out.dtype == np.uint8 and np.array_equal(out, a[:, :, ::-1])            # RGB -> BGR, same size: untouched
    flat = np.full((40, 40, 3), 200, np.uint8)
    assert (cv2_resize_linear_u8(flat, 64, 64) == 200).all()
    # exact 2x decimation in both axes: cv::resize turns INTER_LINEAR into INTER_AREA (2x2 block mean, (sum + 2) >> 2) -- known
    # answers where the bilinear fixed-point kernel would round differently, then product == oracle loops; 2x in ONE axis stays bilinear
    blk = np.array([[0, 1], [1, 0]], np.uint8)
    assert cv2_resize_linear_u8(blk, 1, 1)[0, 0] == 1 and cv2_resize_linear_u8(np.array([[1, 0], [0, 0]], np.uint8), 1, 1)[0, 0] == 0
    assert cv2_resize_linear_u8(np.array([[255, 254], [254, 254]], np.uint8), 1, 1)[0, 0] == 254
    for (h, w) in ((64, 48), (10, 6)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = cv2_resize_linear_u8(a, w // 2, h // 2)
        assert np.array_equal(got, resize_linear_u8(a, (w // 2, h // 2)))
        s = a.astype(np.int64)
        assert np.array_equal(got, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
        assert np.array_equal(cv2_resize_linear_u8(a, w // 2, h), resize_linear_u8(a, (w // 2, h)))


def test_pil_nearest_index_table_is_pillows_rule():
    """pil_nearest_index (kept for Pillow-resampled label images; NOT segment()'s rule) reproduces PIL.Image.resize(..., NEAREST)."""
    import numpy as np
    from PIL import Image
    from oct_segmentation_amd.predict import pil_nearest_index
    for src in (64, 704, 37):
        img = Image.fromarray(np.tile(np.arange(src, dtype=np.float32), (2, 1)))
        for dst in (96, 37, 1000):
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


def test_planner_and_executors_clean_under_asan_ubsan():
    """SURVEY section 5 (sanitizers): the host halves of every source -- csrc/plan.cpp's graph builder, workspace layout, tap tables, launch
    geometry, job tables, and the forward / backward / sliced-backward / optimizer / graph executors -- compiled with
    -fsanitize=address,undefined and driven over every arch x encoder pair the engine builds (32x32, 16 x 704x704, 96x64, 64x160; f32 / bf16 / f16: the plan count asserted below)
    against a recording stand-in for the HIP runtime (tools/hip_host_stubs.cpp: every launch geometry and every memset / copy range is
    checked; nothing runs on a GPU).  `make asan` builds build/asan/plan_dryrun; exit code 0 and the summary line = clean."""
    import subprocess
    csrc = os.path.join(ROOT, 'oct_segmentation_amd', 'csrc')
    b = subprocess.run(['make', '-C', csrc, '-j4', 'asan'], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    r = subprocess.run([os.path.join(csrc, 'build', 'asan', 'plan_dryrun')], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-6000:]
    assert '1296 plans built' in r.stdout    # 18 per pair: 9 archs x 5 ResNets, 4 x 2 RegNetX, 6 x RegNetY-120, 4 / 5 / 4 archs x EfficientNet-B0 / B5 / B7 and '0 errors' in r.stdout, r.stdout
