"""smp's PAN decoder (reference sweep: configs/tune.yaml:18 ``PAN`` -> smp.create_model, src/models/smp/model.py:38-44) at its defaults
(encoder_output_stride 16: layer4 dilated; decoder_channels 32; bilinear align_corners=True), restated in oracle/nets.py.  Engine: the
32-channel branches and the GAU convs on the MFMA kernels, the feature-pyramid-attention block's one-channel pyramid (six conv + BatchNorm2d(1)
+ ReLU layers, two max-pools, three resizes) as one f32 workgroup in csrc/pan.hip, gates / resizes / adds on the existing sweeps.
Bounds as for the other decoders (kink-free nets): logits 1e-4 of scale, Dice 1e-5, every gradient within 2e-3 (or re-judged against
float64), cosine >= 0.999999."""
import pytest
import torch

import test_gpu_net as T
from synth import make_batch

pytestmark = pytest.mark.gpu

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def _oracle_pan(enc, classes, seed):
    m = T._oracle('pan', enc, classes, seed=seed, kinkfree=True)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith('decoder') and name.endswith('conv.bias'):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for mod in [m.decoder.fpa.down1[1], m.decoder.fpa.down2[1], m.decoder.fpa.down3[1], m.decoder.fpa.down3[2], m.decoder.fpa.conv2, m.decoder.fpa.conv1]:
            mod.bn.bias.fill_(8.0)          # one-channel BatchNorms: a bias of -8 would switch the whole pyramid off
        # the kink-free biases (+-8) multiply up through the attention products (pyramid x mid, conv x gate): keep the logits out of saturation,
        # where the Dice gradient is exactly zero
        m.segmentation_head[0].weight.mul_(0.002)
    return m


def _pair(cuda, enc, classes, B, H, W, seed, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_pan(enc, classes, seed).train()
    net = SegNet('pan', enc, classes=classes, device=cuda, compute_dtype=dtype).train()
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(B, classes, max(H, W), seed=seed, empty_last=(classes > 1))
    img, mask = img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    mean, std = torch.tensor(MEAN).view(1, 3, 1, 1), torch.tensor(STD).view(1, 3, 1, 1)
    z = ref((img - mean) / std)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda), normalize=True, mean=MEAN, std=STD)
    torch.cuda.synchronize()
    return ref, net, img, mask, z.detach(), loss_ref, logits.cpu(), loss, stats


@pytest.mark.parametrize('case', [('resnet18', 1, 4, 128, 128), ('resnet34', 2, 3, 128, 192), ('resnet50', 1, 4, 256, 256), ('resnet18', 2, 4, 352, 224)],
                         ids=lambda c: '-'.join(map(str, c)))
def test_pan_train_step_and_every_gradient_fp32(cuda, case):
    from oracle import get_stats
    enc, classes, B, H, W = case
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=9)
    scale, err = z.abs().max().item(), (logits - z).abs().max().item()
    grads = net.named_grads()
    cos, worst, name = T._grad_report(grads, ref)
    print(f'pan/{case}: logits {err:.2e} / {scale:.2f}, loss {abs(loss.item() - loss_ref.item()):.1e}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, scale) and abs(loss.item() - loss_ref.item()) <= 1e-5
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    if bool((z.abs() > 1e-3).all()):
        assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    sd, rd = net.state_dict(), ref.state_dict()
    for k, v in rd.items():
        if k.endswith('running_mean') or k.endswith('running_var'):
            # BatchNorms behind a global pooling see B values per channel: their (unbiased) variance is a difference of nearly equal numbers
            pooled = 'gau' in k and '.conv1.1.bn' in k or '.branch1.1.bn' in k
            assert (sd[k].cpu() - v).abs().max().item() <= (5e-3 if pooled else 1e-4) * max(1.0, v.abs().max().item()), k
    assert cos >= 0.999999
    if worst >= 2e-3:
        from test_gpu_deeplab import judge_gradients
        judge_gradients(ref, grads, img, mask, tag=f'pan/{case}: ', max_rejudged=6)


def test_pan_eval_bf16_and_refusals(cuda):
    from oct_segmentation_amd.engine import SegNet
    ref = _oracle_pan('resnet34', 2, seed=3).eval()
    net = SegNet('pan', 'resnet34', classes=2, device=cuda, compute_dtype=torch.float32).eval()
    net.load_state_dict(ref.state_dict())
    img, _ = make_batch(2, 2, 160, seed=5)
    with torch.no_grad():
        y_ref = ref(img)
    y = net(img.to(cuda), normalize=False).cpu()
    scale, err = y_ref.abs().max().item(), (y - y_ref).abs().max().item()
    print(f'pan/resnet34 eval: logits {err:.2e} / {scale:.2f}')
    assert err <= 1e-4 * max(1.0, scale)
    ref2, net2, img2, mask2, z2, loss_ref, logits, loss, stats = _pair(cuda, 'resnet50', 1, 4, 256, 256, seed=17, dtype=torch.bfloat16)
    cos, worst, name = T._grad_report(net2.named_grads(), ref2)
    s2, e2 = z2.abs().max().item(), (logits - z2).abs().max().item()
    print(f'pan/resnet50 bf16 256^2: logits {e2:.2e}/{s2:.1f} ({e2 / max(s2, 1):.2%}), Dice loss {loss.item():.6f} vs {loss_ref.item():.6f}, grad cosine {cos:.5f}')
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 and cos >= 0.999 and e2 <= 3e-2 * max(1.0, s2)
    with pytest.raises((KeyError, RuntimeError)):
        SegNet('pan', 'efficientnet-b0', classes=1, device=cuda)
    small = SegNet('pan', 'resnet18', classes=1, device=cuda)
    with pytest.raises(RuntimeError):
        small(torch.zeros(1, 3, 64, 64, device=cuda))          # the pyramid pools the 4 x 4 stride-16 feature three times (torch fails there as well)


def test_pan_704_bf16_properties(cuda):
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('pan', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=2).train()
    img, mask = (t.to(cuda) for t in make_batch(9, 1, 704, seed=4))       # 9 x 88 x 88 source pixels of gau1's resize: beyond one grid dimension
    loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=MEAN, std=STD)
    assert torch.isfinite(logits).all() and torch.isfinite(net.arena.grad).all()
    want = DiceLoss()(logits.double().cpu(), mask.double().cpu()).item()
    assert abs(loss.item() - want) <= 2e-6 and int(stats.sum()) == 9 * 704 * 704
    net.eval()
    a = net(img, normalize=True, mean=MEAN, std=STD)
    b = net(img.flip(0), normalize=True, mean=MEAN, std=STD).flip(0)
    assert torch.equal(a, b)
