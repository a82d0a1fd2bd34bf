"""FPN over the ResNet encoders on the engine (first slice): `FPN` is one of the architectures the reference sweeps
(configs/tune.yaml:9-18 -> smp.create_model(arch, ...), src/models/smp/model.py:38-44) and several per-class winners of
eval/tuning/configs_best.xlsx are FPN / resnet nets.  Oracle: oracle/nets.py FPNDecoder (smp 0.3.3 decoders/fpn restated: 1x1 lateral
convs with bias, nearest x2 + add, Conv3x3 + GroupNorm(32) + ReLU + bilinear x2 align_corners, merge 'add', Dropout2d(0.2), 1x1 head +
UpsamplingBilinear2d(4)).  Dropout2d's keep pattern is injected on both sides.  Tolerances as for the other nets (test_gpu_net.py):
fp32 logits 1e-4 of the scale (2e-4 on the 50-layer encoder), Dice 1e-5, counts exact, every parameter gradient on kink-free nets."""
import numpy as np
import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _oracle_fpn(enc, classes, seed, kinkfree):
    from test_gpu_net import _oracle
    m = _oracle('fpn', enc, classes, seed=seed, kinkfree=kinkfree)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith('decoder') and name.endswith('.bias') and p.dim() == 1:        # lateral conv biases and GroupNorm biases
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            if name.startswith('decoder') and '.block.1.weight' in name:                     # GroupNorm weights
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
        if kinkfree:   # GroupNorm pre-activations away from the ReLU kink, as _oracle does for the BatchNorms
            for mod in m.modules():
                if isinstance(mod, torch.nn.GroupNorm):
                    sign = (torch.rand(mod.bias.shape, generator=g) < 0.7).float() * 2 - 1
                    mod.bias.copy_(8.0 * sign)
            # four GroupNorm outputs of ~8 each are summed in front of the head: keep |logits| of order 1 (a saturated sigmoid has no gradient)
            m.segmentation_head[0].weight.mul_(0.03)
    return m


def _pair(cuda, enc, classes, B, H, W, seed, kinkfree, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_fpn(enc, classes, seed, kinkfree)
    net = SegNet('fpn', enc, classes=classes, device=cuda, compute_dtype=dtype)
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())       # smp's module tree, key for key
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(seed + 11)
    img, mask = make_batch(B, classes, max(H, W), seed=seed, empty_last=(classes > 1))
    img, mask = img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    keep = (torch.rand(B, 128, generator=g) < 0.8).float()
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


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 1, 2, 64, 96), ('resnet34', 3, 1, 96, 64), ('resnet50', 2, 2, 64, 64)])
def test_fpn_train_step_parity_fp32(cuda, enc, classes, B, H, W):
    from oracle import get_stats
    from test_gpu_net import _grad_report
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=3, kinkfree=False)
    scale = z.abs().max().item()
    err = (logits - z).abs().max().item()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'fpn/{enc} {classes}c B{B} {H}x{W}: logits {err:.2e} / {scale:.2f}, loss {abs(loss.item() - loss_ref.item()):.1e}, grad cosine {cos:.8f} worst {worst:.1e} ({name})')
    assert err <= (2e-4 if enc == 'resnet50' else 1e-4) * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    away = (z.abs() > 1e-3)
    assert torch.equal((logits > 0)[away], (z > 0)[away])
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    assert cos >= 0.999


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 2, 2, 64, 96), ('resnet50', 1, 2, 64, 64)])
def test_fpn_every_gradient_kinkfree_fp32(cuda, enc, classes, B, H, W):
    """BatchNorm and GroupNorm biases at +-8 (no pre-activation near the ReLU kink): every parameter gradient -- lateral 1x1 convs and
    their biases, 3x3 convs, GroupNorm weights / biases, head, the whole encoder -- within 2e-3 of its largest element, cosine 1 - 1e-6."""
    from test_gpu_net import _grad_report
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=5, kinkfree=True)
    cos, worst, name = _grad_report(net.named_grads(), ref)
    err = (logits - z).abs().max().item()
    print(f'fpn/{enc} kink-free: logits {err:.2e} / {z.abs().max().item():.2f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, z.abs().max().item())
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos >= 0.999999 and worst < 2e-3, (cos, worst, name)


def test_fpn_eval_forward_and_dropout_semantics(cuda):
    """Eval: Dropout2d is the identity, GroupNorm still uses the batch's own statistics (it has no running buffers): logits equal the oracle's
    eval forward.  Training without an injected pattern draws one (kept channels scaled by 1 / 0.8): two steps differ, both finite."""
    from oct_segmentation_amd.engine import SegNet
    ref = _oracle_fpn('resnet18', 2, seed=9, kinkfree=False).eval()
    net = SegNet('fpn', 'resnet18', classes=2, device=cuda, compute_dtype=torch.float32).eval()
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(2, 2, 96, seed=4)
    with torch.no_grad():
        z = ref(img)
    y = net(img.to(cuda)).cpu()
    assert (y - z).abs().max().item() <= 1e-4 * max(1.0, z.abs().max().item())
    net.train()
    torch.manual_seed(1)
    l1, _, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    l2, _, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    assert np.isfinite(l1.item()) and np.isfinite(l2.item()) and l1.item() != l2.item()
    assert torch.isfinite(net.arena.grad).all()


def test_fpn_bf16_704_properties(cuda):
    """BASELINE frame size in bf16 (FPN / resnet50, B = 4): finite, eval forward deterministic and permutation-equivariant, counts
    recounted from the logits, gradient linear in grad_scale; against the fp32 oracle at 256^2: Dice within 1e-3."""
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    net = SegNet('fpn', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=3)
    img, mask = (t.to(cuda) for t in make_batch(4, 1, 704, seed=21))
    net.eval()
    y1 = net(img)
    assert torch.isfinite(y1).all() and torch.equal(y1, net(img))
    perm = torch.tensor([2, 0, 3, 1], device=cuda)
    assert torch.equal(net(img[perm]), y1[perm])
    net.train()
    net.dropout_keep = (torch.rand(4, 128, generator=torch.Generator().manual_seed(2)) < 0.8).float()
    loss, logits, stats = net.train_step_raw(img, mask, grad_scale=1.0)
    g1 = net.arena.grad.clone()
    s = stats.cpu()
    assert int(s[..., 0].sum()) == int(((logits > 0) & (mask > 0)).sum()) and torch.equal(s.sum(-1), torch.full_like(s[..., 0], 704 * 704))
    loss2, _, _ = net.train_step_raw(img, mask, grad_scale=0.5)
    assert abs(loss2.item() - loss.item()) < 1e-6
    ratio = (net.arena.grad.norm() / g1.norm()).item()
    assert abs(ratio - 0.5) < 2e-2
    # bf16 engine vs fp32 oracle
    ref = _oracle_fpn('resnet50', 1, seed=13, kinkfree=True).train()
    net2 = SegNet('fpn', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16)
    net2.load_state_dict(ref.state_dict())
    net2.train()
    im, mk = make_batch(4, 1, 256, seed=17)
    keep = (torch.rand(4, 128, generator=torch.Generator().manual_seed(3)) < 0.8).float()
    ref.decoder.dropout.mask = keep
    net2.dropout_keep = keep
    z = ref(im)
    loss_ref = DiceLoss()(z, mk)
    loss_b, logits_b, _ = net2.train_step_raw(im.to(cuda), mk.to(cuda))
    print(f'fpn/resnet50 bf16 256^2: Dice loss {loss_b.item():.6f} vs {loss_ref.item():.6f}, logits {(logits_b.cpu() - z.detach()).abs().max().item():.2e} / {z.detach().abs().max().item():.1f}')
    assert abs(loss_b.item() - loss_ref.item()) <= 1e-3
