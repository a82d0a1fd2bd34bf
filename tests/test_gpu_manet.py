"""smp's MAnet decoder (reference sweep: configs/tune.yaml:17 ``MAnet`` -> smp.create_model, src/models/smp/model.py:38-44) at its defaults,
restated in oracle/nets.py with both upstream quirks kept (softmax over the WHOLE position map; the attended map reshaped without a
transpose).  Engine: the four PAB convs and every MFAB conv on the MFMA kernels, the attention products / softmax / index map in
csrc/pab.hip, the two squeeze-excite gates of an MFAB (summed behind their sigmoids) in se.hip + effnet.hip's fused excitation, applied
before the nearest-x2 upsample (a per-(image, channel) factor commutes with it).  Bounds as for the other decoders: kink-free nets, logits
1e-4 of scale, Dice 1e-5, every gradient within 2e-3 of its largest element (or re-judged against float64), cosine >= 0.999999."""
import pytest
import torch

import test_gpu_net as T
from synth import make_batch

pytestmark = pytest.mark.gpu

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def _oracle_manet(enc, classes, seed, kinkfree=True):
    m = T._oracle('manet', enc, classes, seed=seed, kinkfree=kinkfree)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for name, p in m.named_parameters():      # biased convs of the PAB and of the SE gates: non-zero biases
            if name.startswith('decoder') and name.endswith('.bias') and ('center' in name or '.SE_' in name):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        # attention logits of a random-init PAB are O(100): a one-hot softmax has no gradient to compare.  Scale the two 1x1 convs so that
        # the position map is spread out
        m.decoder.center.top_conv.weight.mul_(0.05)
        m.decoder.center.center_conv.weight.mul_(0.05)
    return m


def _pair(cuda, enc, classes, B, H, W, seed, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from oracle import DiceLoss
    ref = _oracle_manet(enc, classes, seed).train()
    net = SegNet('manet', enc, classes=classes, device=cuda, compute_dtype=dtype).train()
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


@pytest.mark.parametrize('case', [('resnet18', 1, 2, 64, 64), ('resnet34', 2, 3, 64, 96), ('resnet50', 1, 2, 128, 128), ('resnet18', 2, 2, 256, 192),
                                  ('timm-regnety_120', 1, 2, 64, 64), ('efficientnet-b0', 1, 2, 96, 96)], ids=lambda c: '-'.join(map(str, c)))
def test_manet_train_step_and_every_gradient_fp32(cuda, case):
    from oracle import get_stats
    enc, classes, B, H, W = case
    ref, net, img, mask, z, loss_ref, logits, loss, stats = _pair(cuda, enc, classes, B, H, W, seed=9)
    scale, err = z.abs().max().item(), (logits - z).abs().max().item()
    grads = net.named_grads()
    live = {n for n, p in ref.named_parameters() if p.grad is not None}
    for n, p in ref.named_parameters():          # (EfficientNet keeps its never-run classifier conv: no gradient on either side)
        if p.grad is None:
            assert float(grads[n].abs().max()) == 0.0
            p.grad = torch.zeros_like(p)
    cos, worst, name = T._grad_report(grads, ref)
    print(f'manet/{case}: logits {err:.2e} / {scale:.2f}, loss {abs(loss.item() - loss_ref.item()):.1e}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, scale) and abs(loss.item() - loss_ref.item()) <= 1e-5
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    if bool((z.abs() > 1e-3).all()):
        assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    assert cos >= 0.999999
    if worst >= 2e-3 and not enc.startswith('efficientnet'):
        from test_gpu_deeplab import judge_gradients
        judge_gradients(ref, grads, img, mask, tag=f'manet/{case}: ', max_rejudged=6)
    elif worst >= 2e-3:
        assert worst < 5e-3
    _ = live


def test_manet_eval_and_bf16(cuda):
    from oct_segmentation_amd.engine import SegNet
    ref = _oracle_manet('resnet34', 2, seed=3).eval()
    net = SegNet('manet', 'resnet34', classes=2, device=cuda, compute_dtype=torch.float32).eval()
    net.load_state_dict(ref.state_dict())
    img, _ = make_batch(2, 2, 128, seed=5)
    with torch.no_grad():
        y_ref = ref(img)
    y = net(img.to(cuda), normalize=False).cpu()
    scale, err = y_ref.abs().max().item(), (y - y_ref).abs().max().item()
    print(f'manet/resnet34 eval: logits {err:.2e} / {scale:.2f}')
    assert err <= 1e-4 * max(1.0, scale)
    half = SegNet('manet', 'resnet34', classes=2, device=cuda, compute_dtype=torch.float16).eval()
    half.load_state_dict(ref.state_dict())
    assert torch.isfinite(half(img.to(cuda), normalize=False)).all()
    ref2, net2, img2, mask2, z2, loss_ref, logits, loss, stats = _pair(cuda, 'resnet50', 1, 2, 256, 256, seed=17, dtype=torch.bfloat16)
    cos, worst, name = T._grad_report(net2.named_grads(), ref2)
    s2, e2 = z2.abs().max().item(), (logits - z2).abs().max().item()
    print(f'manet/resnet50 bf16 256^2: logits {e2:.2e}/{s2:.1f} ({e2 / max(s2, 1):.2%}), Dice loss {loss.item():.6f} vs {loss_ref.item():.6f}, grad cosine {cos:.5f}')
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 and cos >= 0.999 and e2 <= 3e-2 * max(1.0, s2)


def test_manet_704_bf16_properties(cuda):
    """BASELINE frame size: the position map of the 22 x 22 feature has 484^2 entries per frame."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('manet', 'resnet50', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=2).train()
    img, mask = (t.to(cuda) for t in make_batch(2, 1, 704, seed=4))
    loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=MEAN, std=STD)
    assert torch.isfinite(logits).all() and torch.isfinite(net.arena.grad).all()
    want = DiceLoss()(logits.double().cpu(), mask.double().cpu()).item()
    assert abs(loss.item() - want) <= 2e-6 and int(stats.sum()) == 2 * 704 * 704
    net.eval()
    a = net(img, normalize=True, mean=MEAN, std=STD)
    b = net(img.flip(0), normalize=True, mean=MEAN, std=STD).flip(0)
    assert torch.equal(a, b)
