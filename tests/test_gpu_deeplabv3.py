"""DeepLabV3 over the ResNet encoders on the engine: `DeepLabV3` is one of the architectures the reference sweeps
(configs/tune.yaml:9-18 -> smp.create_model(arch, ...), src/models/smp/model.py:38-44; DeepLabV3/resnet50-101 are per-class winners in
eval/tuning/configs_best.xlsx).  Oracle: oracle/nets.py DeepLabV3Decoder + ResNetEncoder.make_dilated(8) (smp 0.3.3 restated: layer3 at
dilation 2, layer4 at dilation 4, DENSE ASPP = 1x1 + three dilated 3x3 (12, 24, 36) + image pooling, 1x1 project + BN + ReLU +
Dropout(0.5), 3x3 conv + BN + ReLU, 1x1 head + UpsamplingBilinear2d(8)).  The engine has no dilated conv kernel: layer3 / layer4 run on
nested parity re-arrangements and each dense dilated conv runs as a plain 3x3 conv on a mosaic of its rate^2 sub-grids (deeplab.hip), its
BatchNorm statistics taken from the un-mosaicked tensor -- this file is the test of both identities.  Tolerances as for the other nets."""
import numpy as np
import pytest
import torch

from synth import make_batch
from test_gpu_deeplab import judge_gradients

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _oracle_dl3(enc, classes, seed, kinkfree):
    from test_gpu_net import _oracle
    m = _oracle('deeplabv3', enc, classes, seed=seed, kinkfree=kinkfree)
    if kinkfree:
        with torch.no_grad():
            m.segmentation_head[0].weight.mul_(0.03)
    return m


def _pair(cuda, enc, classes, B, H, W, seed, kinkfree, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_dl3(enc, classes, seed, kinkfree)
    net = SegNet('deeplabv3', enc, classes=classes, device=cuda, compute_dtype=dtype)
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(seed + 11)
    img, mask = make_batch(B, classes, max(H, W), seed=seed, empty_last=(classes > 1))
    img, mask = img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    img = (img * (0.35 + 0.65 * torch.arange(B).view(B, 1, 1, 1) / max(1, B - 1))).round().contiguous()   # (pooled-branch BatchNorm: see test_gpu_deeplab)
    keep = (torch.rand(B, 256, H // 8, W // 8, generator=g) < 0.5).float()
    ref.train(); net.train()
    ref.decoder.dropout.mask = keep
    net.dropout_keep = keep
    mean = torch.tensor(MEAN).view(1, 3, 1, 1); std = torch.tensor(STD).view(1, 3, 1, 1)
    z = ref((img - mean) / std)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda), normalize=True, mean=MEAN, std=STD)
    torch.cuda.synchronize()
    return ref, net, img, mask, z.detach(), loss_ref, logits.cpu(), loss, stats


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 1, 2, 64, 96), ('resnet34', 3, 3, 96, 64), ('resnet50', 2, 4, 64, 64)])
def test_deeplabv3_train_step_parity_fp32(cuda, enc, classes, B, H, W):
    from oracle import get_stats
    from test_gpu_net import _grad_report
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=3, kinkfree=False)
    scale = z.abs().max().item()
    err = (logits - z).abs().max().item()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'deeplabv3/{enc} {classes}c B{B} {H}x{W}: logits {err:.2e} / {scale:.2f}, loss {abs(loss.item() - loss_ref.item()):.1e}, grad cosine {cos:.8f} worst {worst:.1e} ({name})')
    assert err <= (2e-4 if enc == 'resnet50' else 1e-4) * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    away = (z.abs() > 1e-3)
    assert torch.equal((logits > 0)[away], (z > 0)[away])
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    assert cos >= (0.998 if enc == 'resnet50' else 0.999)
    sd, rd = net.state_dict(), ref.state_dict()
    for k in rd:
        if k.endswith('running_mean') or k.endswith('running_var'):
            d = (sd[k].cpu() - rd[k]).abs().max().item()
            assert d <= 1e-4 * max(1.0, rd[k].abs().max().item()), (k, d)


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 2, 4, 64, 96), ('resnet50', 1, 4, 64, 64), ('resnet18', 1, 2, 352, 352)])
def test_deeplabv3_every_gradient_kinkfree_fp32(cuda, enc, classes, B, H, W):
    """BatchNorm biases at +-8: every parameter gradient -- the dense dilated convs through their mosaics, the nested-parity layer3 /
    layer4, the pooled branch -- within 2e-3 of its largest element or re-judged against float64.  The 352^2 case has a 44 x 44 stride-8
    map: every tap of the rates 12 / 24 / 36 lands inside it (on the small frames only centre taps touch data)."""
    from test_gpu_net import _grad_report
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=5, kinkfree=True)
    grads = net.named_grads()
    cos, worst, name = _grad_report(grads, ref)
    err = (logits - z).abs().max().item()
    print(f'deeplabv3/{enc} B{B} {H}x{W} kink-free: logits {err:.2e} / {z.abs().max().item():.2f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, z.abs().max().item())
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos >= 0.999999
    judge_gradients(ref, grads, img, mask, max_rejudged=6)


def test_deeplabv3_eval_bf16_and_704(cuda):
    """Eval equals the oracle (BatchNorm of the mosaic convs folded into their weights); BASELINE frame size in bf16: finite, deterministic,
    counts recounted from the logits, gradient linear in grad_scale; bf16 vs the fp32 oracle at 256^2: Dice within 1e-3; f16 eval finite."""
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_dl3('resnet18', 2, seed=9, kinkfree=False).eval()
    net = SegNet('deeplabv3', 'resnet18', classes=2, device=cuda, compute_dtype=torch.float32).eval()
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(2, 2, 96, seed=4)
    with torch.no_grad():
        z = ref(img)
    y = net(img.to(cuda)).cpu()
    assert (y - z).abs().max().item() <= 1e-4 * max(1.0, z.abs().max().item())
    big = SegNet('deeplabv3', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=3)
    im, mk = (t.to(cuda) for t in make_batch(2, 1, 704, seed=21))
    big.eval()
    y1 = big(im)
    assert torch.isfinite(y1).all() and torch.equal(y1, big(im))
    big.train()
    big.dropout_keep = (torch.rand(2, 88, 88, 256, generator=torch.Generator().manual_seed(2)) < 0.5).float()
    loss, logits, stats = big.train_step_raw(im, mk, grad_scale=1.0)
    g1 = big.arena.grad.clone()
    s = stats.cpu()
    assert int(s[..., 0].sum()) == int(((logits > 0) & (mk > 0)).sum()) and torch.equal(s.sum(-1), torch.full_like(s[..., 0], 704 * 704))
    loss2, _, _ = big.train_step_raw(im, mk, grad_scale=0.5)
    assert abs(loss2.item() - loss.item()) < 1e-6 and abs((big.arena.grad.norm() / g1.norm()).item() - 0.5) < 2e-2
    ref2 = _oracle_dl3('resnet50', 1, seed=13, kinkfree=True).train()
    net2 = SegNet('deeplabv3', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16)
    net2.load_state_dict(ref2.state_dict())
    net2.train()
    i2, m2 = make_batch(4, 1, 256, seed=17)
    keep = (torch.rand(4, 256, 32, 32, generator=torch.Generator().manual_seed(3)) < 0.5).float()
    ref2.decoder.dropout.mask = keep
    net2.dropout_keep = keep
    z2 = ref2(i2)
    loss_ref = DiceLoss()(z2, m2)
    loss_b, logits_b, _ = net2.train_step_raw(i2.to(cuda), m2.to(cuda))
    print(f'deeplabv3/resnet50 bf16 256^2: Dice loss {loss_b.item():.6f} vs {loss_ref.item():.6f}, logits {(logits_b.cpu() - z2.detach()).abs().max().item():.2e} / {z2.detach().abs().max().item():.1f}')
    assert abs(loss_b.item() - loss_ref.item()) <= 1e-3
    half = SegNet('deeplabv3', 'resnet50', classes=1, device=cuda, compute_dtype=torch.float16).eval()
    half.load_state_dict(net2.state_dict())
    assert torch.isfinite(half(i2.to(cuda))).all() and np.isfinite(loss_b.item())
