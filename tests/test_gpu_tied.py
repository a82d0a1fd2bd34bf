"""The tied decomposition of the decoder's (nearest x2, concat, 3x3) layers -- smp's DecoderBlock, F.interpolate(scale_factor=2, mode='nearest')
+ torch.cat + Conv2dReLU (reference graph: src/models/smp/model.py:65-71 builds it through smp.create_model) -- must be the SAME function of the
same weights as the plain 3x3 path: over the upsampled source's channels the 3x3 equals a ConvTranspose2d(k4, s2, p1) of the low-resolution map
whose 4x4 kernel sums the 3x3 taps that land on one source pixel (plan.h ConvLayer::tie).  Exact up to summation order, so:

  * the identity itself and the fold of the 4x4 kernel's gradient are pinned in float64 on the CPU (tests/test_tied_identity.py);
  * each pass on its own against an untied plan of the same network and batch, so that a wrong forward cannot hide behind a matching wrong
    gradient.  `w`: nothing but the tied layers' weight gradients may move, and those only by the order of an fp32 sum (the same bf16 products).
    `d`: the forward is bit-identical, so the ReLU masks are too and the backward is the same linear map with the 4x4 image rounded once
    instead of tap by tap -- every parameter's gradient keeps its direction to 5e-4.  `f`: the deepest tied layer sees bit-identical inputs, its
    raw output is compared element-wise at bf16 rounding (the skip launch stores a rounded partial sum the parity launches add to: two roundings
    more than the plain path's one); loss (north_star: 1e-3), logits and running statistics of the whole step follow;
  * all passes together (OCTSEG_TIED=fdw): loss, logits, the global gradient direction;
  * the executed multiply-accumulate count the plan reports drops by 5/9 of the tied share and never for the reference graph's count.
"""
import os

import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _step(cuda, arch, enc, tied, S=128, B=2, classes=2, dtype=torch.bfloat16, probe=None):
    from oct_segmentation_amd.engine import SegNet, debug_tensor
    old = os.environ.pop('OCTSEG_TIED', None)
    os.environ['OCTSEG_TIED'] = tied or '0'          # ('' = the plain path: the library's default ties both gradients)
    try:
        net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=dtype, seed=5).train()
        img, mask = (t.to(cuda) for t in make_batch(B, classes, S, seed=9))
        loss, logits, stats = net.train_step_raw(img, mask, True, MEAN, STD)
        torch.cuda.synchronize()
        out = {'loss': float(loss), 'logits': logits.float().clone(), 'grads': net.named_grads(),
               'bufs': {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k},
               'alg': net.fwd_macs(B, S, S), 'exec': net.exec_macs(B, S, S)}
        if probe:
            out['probe'] = debug_tensor(net, net._plan(B, S, S), probe)
        return out
    finally:
        os.environ.pop('OCTSEG_TIED', None)
        if old is not None:
            os.environ['OCTSEG_TIED'] = old


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


# (deepest tied layer of the decoder: its sources come straight from the encoder)
CASES = [('unet', 'resnet18', 'decoder.blocks.0.conv1.0'), ('unetplusplus', 'resnet34', 'decoder.blocks.x_0_0.conv1.0'),
         ('unet', 'resnet50', 'decoder.blocks.0.conv1.0'), ('manet', 'resnet18', 'decoder.blocks.0.conv1.0')]


def _check_macs(plain, tied_run, tied):
    assert tied_run['alg'] == plain['alg'] and plain['exec'] == (plain['alg'],) * 3          # the reference graph's count never moves
    for k, c in enumerate('fdw'):
        assert (tied_run['exec'][k] < tied_run['alg']) == (c in tied), f'executed MACs of pass {c} with OCTSEG_TIED={tied}: {tied_run["exec"]}'


@pytest.mark.parametrize('arch,enc,probe', CASES, ids=lambda v: str(v))
def test_tied_weight_gradient(cuda, arch, enc, probe):
    a, b = _step(cuda, arch, enc, ''), _step(cuda, arch, enc, 'w')
    _check_macs(a, b, 'w')
    assert b['loss'] == a['loss'] and torch.equal(b['logits'], a['logits'])
    tied_keys = 0
    for k, g0 in a['grads'].items():
        g1 = b['grads'][k]
        if k.endswith('conv1.0.weight') and 'decoder.blocks' in k and not torch.equal(g0, g1):
            tied_keys += 1
        # fp32 atomics: the plain path's own run-to-run spread is a few ulp of the largest partial sum
        assert torch.allclose(g1, g0, rtol=2e-3, atol=2e-5 * float(g0.abs().max()) + 1e-30), \
            f'{k}: weight gradient moved by {float((g1 - g0).abs().max())} of {float(g0.abs().max())}'
        assert _cos(g0, g1) >= 0.999999 or float(g0.norm()) == 0, f'{k}: cosine {_cos(g0, g1)}'
    assert tied_keys >= 1, 'no tied layer in this network: the case tests nothing'


@pytest.mark.parametrize('arch,enc,probe', CASES, ids=lambda v: str(v))
def test_tied_data_gradient(cuda, arch, enc, probe):
    a, b = _step(cuda, arch, enc, ''), _step(cuda, arch, enc, 'd')
    _check_macs(a, b, 'd')
    assert b['loss'] == a['loss'] and torch.equal(b['logits'], a['logits'])
    moved = 0
    for k, g0 in a['grads'].items():
        g1 = b['grads'][k]
        moved += int(not torch.equal(g0, g1))
        if float(g0.norm()) > 0 and g0.numel() >= 16:
            # (MAnet's PAB: the gradient through its global softmax is a difference of large terms; its biases keep 0.9993)
            floor = 0.995 if k.startswith('decoder.center.') else 0.9995
            assert _cos(g0, g1) >= floor, f'{arch}/{enc}: gradient of {k} has cosine {_cos(g0, g1)} against the plain path'
            r = float(g1.norm() / g0.norm())          # (MAnet's PAB: the gradient through its global softmax is a difference of large terms -- 0.975)
            assert 0.95 <= r <= 1.05, f'{k}: gradient norm ratio {r}'
    assert moved >= 1


@pytest.mark.parametrize('arch,enc,probe', CASES, ids=lambda v: str(v))
def test_tied_forward(cuda, arch, enc, probe):
    a, b = _step(cuda, arch, enc, '', probe=probe), _step(cuda, arch, enc, 'f', probe=probe)
    _check_macs(a, b, 'f')
    y0, y1 = a['probe'], b['probe']
    assert y0.shape == y1.shape and not torch.equal(y0, y1)
    scale = float(y0.abs().max())
    err = (y1 - y0).abs()
    # three bf16 roundings at most (2^-8 relative each) on sums whose partial terms can exceed the result: bound by the tensor's scale
    assert float(err.max()) <= 3 * 2.0 ** -8 * scale, f'{probe}: raw output differs by {float(err.max())} of {scale}'
    assert float((y1 - y0).norm() / y0.norm()) <= 6e-3, f'{probe}: relative L2 error {float((y1 - y0).norm() / y0.norm())}'
    assert abs(b['loss'] - a['loss']) <= 1e-3, f'loss {b["loss"]} against {a["loss"]}'          # north_star's Dice bound for bf16
    zs = float(a['logits'].abs().max())
    assert float((b['logits'] - a['logits']).abs().max()) <= 0.05 * zs
    for k in a['bufs']:
        assert torch.allclose(b['bufs'][k], a['bufs'][k], rtol=2e-2, atol=2e-3), f'{k}: running statistic moved by {float((b["bufs"][k] - a["bufs"][k]).abs().max())}'


@pytest.mark.parametrize('arch,enc,probe', CASES[:2], ids=lambda v: str(v))
def test_tied_all_passes(cuda, arch, enc, probe):
    a, b = _step(cuda, arch, enc, ''), _step(cuda, arch, enc, 'fdw')
    _check_macs(a, b, 'fdw')
    assert abs(b['loss'] - a['loss']) <= 1e-3
    zs = float(a['logits'].abs().max())
    assert float((b['logits'] - a['logits']).abs().max()) <= 0.05 * zs
    f0 = torch.cat([g.flatten() for g in a['grads'].values()]); f1 = torch.cat([b['grads'][k].flatten() for k in a['grads']])
    # (a perturbed forward moves ReLU masks: the same spread the bf16 engine shows against the fp32 oracle away from kink-free inputs)
    assert _cos(f0, f1) >= 0.97, f'global gradient cosine {_cos(f0, f1)}'
    assert torch.isfinite(f1).all()


@pytest.mark.parametrize('tied', ['d', 'w'])
def test_tied_gradients_on_ragged_maps(cuda, tied):
    """96 x 96 frames, three of them: decoder maps of 6, 12, 24, 48 pixels a side (3 x 3 low-resolution sources, tiles of 8 x 16 and 16 x 16 pixels
    filled to a fraction, an odd batch) through the fused ConvTranspose2d weight gradient and the masked parity-plane data gradient."""
    a = _step(cuda, 'unetplusplus', 'resnet18', '', S=96, B=3)
    b = _step(cuda, 'unetplusplus', 'resnet18', tied, S=96, B=3)
    _check_macs(a, b, tied)
    assert b['loss'] == a['loss'] and torch.equal(b['logits'], a['logits'])
    for k, g0 in a['grads'].items():
        g1 = b['grads'][k]
        if float(g0.norm()) > 0 and g0.numel() >= 16:
            assert _cos(g0, g1) >= (0.999999 if tied == 'w' else 0.9995), f'{k}: cosine {_cos(g0, g1)} with OCTSEG_TIED={tied}'


def test_tied_default_is_the_two_gradients(cuda):
    from oct_segmentation_amd.engine import SegNet
    old = os.environ.pop('OCTSEG_TIED', None)
    try:
        net = SegNet('unet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=1)
        alg, ex = net.fwd_macs(2, 64, 64), net.exec_macs(2, 64, 64)
        assert ex[0] == alg and ex[1] < alg and ex[2] < alg
    finally:
        if old is not None:
            os.environ['OCTSEG_TIED'] = old


def test_tied_plan_needs_two_byte_dtype(cuda):
    """fp32 plans keep the plain path (their 1e-4 parity bound is stated against the reference's summation, tap by tap)."""
    r = _step(cuda, 'unet', 'resnet18', 'fdw', dtype=torch.float32)
    assert r['exec'] == (r['alg'],) * 3
