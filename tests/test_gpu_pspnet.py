"""PSPNet over the ResNet encoders on the engine: `PSPNet` is one of the architectures the reference sweeps
(configs/tune.yaml:9-18 -> smp.create_model(arch, ...), src/models/smp/model.py:38-44).  Oracle: oracle/nets.py PSPDecoder (smp 0.3.3
restated: encoder_depth 3 -- layer3 / layer4 stay in the module and in state_dict but never run --, adaptive average pooling to 1 / 2 /
3 / 6 bins (overlapping bins where the size does not divide), 1x1 conv + BatchNorm + ReLU per bin size (a biased conv without BatchNorm
for the 1x1 bin), bilinear resize back with align_corners=True, concat with the feature, 1x1 conv to 512 + BatchNorm + ReLU, Dropout2d(0.2),
3x3 head + UpsamplingBilinear2d(8)).  The Dropout2d pattern is injected on both sides.  Tolerances as for the other nets."""
import numpy as np
import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _oracle_psp(enc, classes, seed, kinkfree):
    from test_gpu_net import _oracle
    m = _oracle('pspnet', enc, classes, seed=seed, kinkfree=kinkfree)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        m.decoder.psp.blocks[0].pool[1][0].bias.copy_(0.3 * torch.randn(m.decoder.psp.blocks[0].pool[1][0].bias.shape, generator=g))
        if kinkfree:
            m.segmentation_head[0].weight.mul_(0.03)
    return m


def _report(grads, ref):
    """test_gpu_net._grad_report over the parameters that have a gradient; the encoder stages behind the last feature must stay at zero."""
    live = [(n, p) for n, p in ref.named_parameters() if p.grad is not None]
    dead = [n for n, p in ref.named_parameters() if p.grad is None]
    assert dead and all(n.startswith('encoder.layer3') or n.startswith('encoder.layer4') for n in dead)
    for n in dead:
        assert float(grads[n].abs().max()) == 0.0, n
    gmax = max(p.grad.abs().max().item() for _, p in live)
    num = da = db = 0.0
    worst, worst_name = 0.0, ''
    for n, p in live:
        a, b = grads[n].cpu().double(), p.grad.double()
        num += float((a * b).sum()); da += float((a * a).sum()); db += float((b * b).sum())
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3 * gmax)
        if e > worst:
            worst, worst_name = e, n
    return num / (da ** 0.5 * db ** 0.5 + 1e-30), worst, worst_name


def _pair(cuda, enc, classes, B, H, W, seed, kinkfree, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_psp(enc, classes, seed, kinkfree)
    net = SegNet('pspnet', enc, classes=classes, device=cuda, compute_dtype=dtype)
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())       # smp's module tree, key for key (layer3 / layer4 included)
    net.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(seed + 11)
    img, mask = make_batch(B, classes, max(H, W), seed=seed, empty_last=(classes > 1))
    img, mask = img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    img = (img * (0.35 + 0.65 * torch.arange(B).view(B, 1, 1, 1) / max(1, B - 1))).round().contiguous()   # (small-count BatchNorms: see test_gpu_deeplab)
    keep = (torch.rand(B, 512, generator=g) < 0.8).float()
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


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 1, 2, 64, 96), ('resnet34', 3, 3, 160, 96), ('resnet50', 2, 4, 96, 128)])
def test_pspnet_train_step_parity_fp32(cuda, enc, classes, B, H, W):
    from oracle import get_stats
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=3, kinkfree=False)
    scale = z.abs().max().item()
    err = (logits - z).abs().max().item()
    cos, worst, name = _report(net.named_grads(), ref)
    print(f'pspnet/{enc} {classes}c B{B} {H}x{W}: logits {err:.2e} / {scale:.2f}, loss {abs(loss.item() - loss_ref.item()):.1e}, grad cosine {cos:.8f} worst {worst:.1e} ({name})')
    assert err <= (2e-4 if enc == 'resnet50' else 1e-4) * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    away = (z.abs() > 1e-3)
    assert torch.equal((logits > 0)[away], (z > 0)[away])
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    assert cos >= 0.999
    sd, rd = net.state_dict(), ref.state_dict()
    for k in rd:
        if k.endswith('running_mean') or k.endswith('running_var'):
            d = (sd[k].cpu() - rd[k]).abs().max().item()
            assert d <= 1e-4 * max(1.0, rd[k].abs().max().item()), (k, d)      # (the never-run stages keep their buffers untouched)


@pytest.mark.parametrize('enc,classes,B,H,W', [('resnet18', 2, 3, 96, 160), ('resnet50', 1, 4, 64, 96)])
def test_pspnet_every_gradient_kinkfree_fp32(cuda, enc, classes, B, H, W):
    """BatchNorm biases at +-8: every parameter gradient that exists -- bin convs, the biased 1x1-bin conv, the fuse conv, the encoder
    up to layer2 -- within 2e-3 of its largest element (overlapping bins: 12 x 20 and 8 x 12 maps into 3 and 6 bins)."""
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=5, kinkfree=True)
    cos, worst, name = _report(net.named_grads(), ref)
    err = (logits - z).abs().max().item()
    print(f'pspnet/{enc} kink-free: logits {err:.2e} / {z.abs().max().item():.2f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, z.abs().max().item())
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos >= 0.999999 and worst < 2e-3, (cos, worst, name)


def test_pspnet_eval_bf16_and_704(cuda):
    """Eval (dropout = identity, running statistics) equals the oracle; BASELINE frame size in bf16: finite, deterministic, counts
    recounted from the logits, gradient linear in grad_scale; bf16 vs the fp32 oracle at 256^2: Dice within 1e-3; f16 eval finite."""
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_psp('resnet18', 2, seed=9, kinkfree=False).eval()
    net = SegNet('pspnet', 'resnet18', classes=2, device=cuda, compute_dtype=torch.float32).eval()
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(2, 2, 96, seed=4)
    with torch.no_grad():
        z = ref(img)
    y = net(img.to(cuda)).cpu()
    assert (y - z).abs().max().item() <= 1e-4 * max(1.0, z.abs().max().item())
    big = SegNet('pspnet', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=3)
    im, mk = (t.to(cuda) for t in make_batch(4, 1, 704, seed=21))
    big.eval()
    y1 = big(im)
    assert torch.isfinite(y1).all() and torch.equal(y1, big(im))
    big.train()
    big.dropout_keep = (torch.rand(4, 512, generator=torch.Generator().manual_seed(2)) < 0.8).float()
    loss, logits, stats = big.train_step_raw(im, mk, grad_scale=1.0)
    g1 = big.arena.grad.clone()
    s = stats.cpu()
    assert int(s[..., 0].sum()) == int(((logits > 0) & (mk > 0)).sum()) and torch.equal(s.sum(-1), torch.full_like(s[..., 0], 704 * 704))
    loss2, _, _ = big.train_step_raw(im, mk, grad_scale=0.5)
    assert abs(loss2.item() - loss.item()) < 1e-6 and abs((big.arena.grad.norm() / g1.norm()).item() - 0.5) < 2e-2
    ref2 = _oracle_psp('resnet50', 1, seed=13, kinkfree=True).train()
    net2 = SegNet('pspnet', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16)
    net2.load_state_dict(ref2.state_dict())
    net2.train()
    i2, m2 = make_batch(4, 1, 256, seed=17)
    keep = (torch.rand(4, 512, generator=torch.Generator().manual_seed(3)) < 0.8).float()
    ref2.decoder.dropout.mask = keep
    net2.dropout_keep = keep
    z2 = ref2(i2)
    loss_ref = DiceLoss()(z2, m2)
    loss_b, logits_b, _ = net2.train_step_raw(i2.to(cuda), m2.to(cuda))
    print(f'pspnet/resnet50 bf16 256^2: Dice loss {loss_b.item():.6f} vs {loss_ref.item():.6f}, logits {(logits_b.cpu() - z2.detach()).abs().max().item():.2e} / {z2.detach().abs().max().item():.1f}')
    assert abs(loss_b.item() - loss_ref.item()) <= 1e-3
    half = SegNet('pspnet', 'resnet50', classes=1, device=cuda, compute_dtype=torch.float16).eval()
    half.load_state_dict(net2.state_dict())
    assert torch.isfinite(half(i2.to(cuda))).all()
    assert np.isfinite(loss_b.item())


def test_pspnet_dead_stages_are_left_alone_by_the_optimizer(cuda):
    """torch optimizers skip parameters without a gradient (encoder.layer3 / layer4 of smp's encoder_depth=3 PSPNet), so their weight decay
    never touches them; the fused optimizer steps over the live ranges of the arena only.  Live parameters follow torch.optim.Adam."""
    from oct_segmentation_amd.engine import SegNet
    from oct_segmentation_amd.model import FusedOptimizer
    from oracle import DiceLoss
    ref = _oracle_psp('resnet18', 1, seed=21, kinkfree=False).train()
    net = SegNet('pspnet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.float32).train()
    net.load_state_dict(ref.state_dict())
    assert len(net.live_ranges) == 2
    img, mask = make_batch(3, 1, 64, seed=8)
    keep = (torch.rand(3, 512, generator=torch.Generator().manual_seed(1)) < 0.8).float()
    ref.decoder.dropout.mask = keep
    net.dropout_keep = keep
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-2)
    opt = FusedOptimizer(net, 'Adam', 1e-3, 1e-2)
    before = {k: v.clone() for k, v in ref.state_dict().items()}
    DiceLoss()(ref(img), mask).backward()
    opt_ref.step()
    net.train_step_raw(img.to(cuda), mask.to(cuda))
    opt.step()
    sd, rd = net.state_dict(), ref.state_dict()
    for k, v in rd.items():
        if k.startswith('encoder.layer3.') or k.startswith('encoder.layer4.'):
            assert torch.equal(sd[k].cpu(), before[k]), k                    # untouched, as in torch
        elif v.dtype.is_floating_point and 'running' not in k:
            d = (sd[k].cpu() - v).abs()
            # Adam's first step is lr * g / (|g| + eps): an element whose gradient is rounding noise may move by lr in either direction
            assert d.max().item() <= 2.2e-3 and d.mean().item() <= 5e-5, (k, d.max().item(), d.mean().item())


def test_pspnet_dead_stages_under_a_torch_optimizer(cuda):
    """with fused_optimizer=False the arena is ONE nn.Parameter whose gradient is zero over the never-run stages, so torch's
    weight decay would still move them (in the reference their grad is None and the optimizer skips them).  configure_optimizers wraps
    the torch optimizer so that the dead ranges come back bit-identical after every step, for every optimizer kind."""
    from oct_segmentation_amd.model import OCTSegmentationModel
    img, mask = make_batch(3, 1, 64, seed=8)
    for name in ('Adam', 'SGD', 'RMSprop', 'RAdam'):
        m = OCTSegmentationModel('PSPNet', 'resnet18', 'p', 3, ['Lumen'], lr=1e-3, weight_decay=1e-2, optimizer_name=name, device=cuda,
                                 compute_dtype=torch.float32, fused_optimizer=False)
        m.train()
        before = m.model.state_dict()
        opt = m.configure_optimizers()
        for _ in range(2):
            opt.zero_grad()
            m.training_step((img.to(cuda), mask.to(cuda)))['loss'].backward()
            opt.step()
        after = m.model.state_dict()
        moved = 0
        for k, v in before.items():
            if k.startswith('encoder.layer3.') or k.startswith('encoder.layer4.'):
                assert torch.equal(after[k], v), (name, k)
            elif v.dtype.is_floating_point and 'running' not in k and k.endswith('weight'):
                moved += int(not torch.equal(after[k], v))
        assert moved > 10, name          # the live parameters did step
