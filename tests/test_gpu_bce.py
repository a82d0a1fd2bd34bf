"""BCE loss + gradient (north_star: "the Dice/BCE loss+grad"; the reference builds DiceLoss only, src/models/smp/model.py:55, so the
switch defaults to Dice).  ``loss='bce'`` = torch.nn.functional.binary_cross_entropy_with_logits (mean), ``'dice+bce'`` = their sum,
both out of the one fused pass behind octseg_dice_forward and differentiated by the backward's dL/dlogits kernel.
Bounds: loss <= 1e-5, every-parameter gradient cosine >= 0.999999 on kink-free nets, a saturated-logit case, the facade switch,
non-contiguous frames (the stem's weight gradient re-reads the frame in the backward)."""
import pytest
import torch

from synth import make_batch
from test_gpu_net import _grad_report, _oracle

pytestmark = pytest.mark.gpu

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def _oracle_loss(kind):
    from oracle import DiceBCELoss, DiceLoss, bce_with_logits
    return {'dice': DiceLoss(), 'bce': bce_with_logits, 'dice+bce': DiceBCELoss()}[kind]


@pytest.mark.parametrize('kind', ['bce', 'dice+bce'])
@pytest.mark.parametrize('cfg', [('unet', 'resnet18', 1, 2, 64), ('linknet', 'resnet18', 2, 3, 64), ('unetplusplus', 'resnet34', 2, 2, 96)],
                         ids=lambda c: '-'.join(map(str, c)))
def test_bce_train_step_parity_fp32(cuda, cfg, kind):
    from oct_segmentation_amd.engine import SegNet
    arch, enc, classes, B, S = cfg
    mean, std = torch.tensor(MEAN).view(1, 3, 1, 1), torch.tensor(STD).view(1, 3, 1, 1)
    ref = _oracle(arch, enc, classes, kinkfree=True)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32, loss=kind)
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(B, classes, S, seed=21, empty_last=(classes > 1))
    ref.train(); net.train()
    z = ref((img - mean) / std)
    loss_ref = _oracle_loss(kind)(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda), normalize=True, mean=MEAN, std=STD)
    torch.cuda.synchronize()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'{cfg} {kind}: loss {loss.item():.7f} vs {loss_ref.item():.7f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, abs(loss_ref.item()))
    assert cos >= 0.999999 and worst <= 2e-3
    # the counts do not depend on the criterion
    from oracle import get_stats
    tp, fp, fn, tn = get_stats((logits.cpu().sigmoid() > 0.5).long(), mask.long())
    assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))


def test_bce_of_saturated_logits(cuda):
    """|z| up to ~25 in both signs: the stable form max(-z, 0) + (1 - t) z + log1p(exp(-|z|)) must neither overflow nor lose the
    linear tail, and sigmoid(z) - t must keep exp(-|z|)-sized gradients of confidently right pixels."""
    from oct_segmentation_amd.engine import SegNet
    ref = _oracle('unet', 'resnet18', 1, kinkfree=True)
    img, mask = make_batch(3, 1, 64, seed=19)
    ref.train()
    with torch.no_grad():
        z0 = ref(img)
        lo, hi = z0.quantile(0.2), z0.quantile(0.8)
        k = 50.0 / (hi - lo)
        ref.segmentation_head[0].bias.sub_((hi + lo) / 2)
        ref.segmentation_head[0].weight.mul_(k)
        ref.segmentation_head[0].bias.mul_(k)
    net = SegNet('unet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.float32, loss='bce')
    net.load_state_dict(ref.state_dict())
    net.train()
    z = ref(img)
    assert float((z > 17).float().mean()) > 0.1 and float((z < -17).float().mean()) > 0.1
    from oracle import bce_with_logits
    loss_ref = bce_with_logits(z, mask)
    loss_ref.backward()
    loss, _, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'saturated bce: loss {loss.item():.6f} vs {loss_ref.item():.6f}, cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * max(1.0, loss_ref.item())
    assert cos >= 0.999999 and worst <= 5e-3


def test_loss_switch_of_the_facade_and_default(cuda):
    """OCTSegmentationModel(loss=...) reaches the plans; the default stays the reference's Dice; switching the criterion of a live net
    re-targets its existing plans; an unknown name raises."""
    from oct_segmentation_amd.model import OCTSegmentationModel
    from oracle import DiceLoss, bce_with_logits
    img, mask = make_batch(2, 1, 64, seed=5)
    m = OCTSegmentationModel('Unet', 'resnet18', 'u', 3, ['Lumen'], device=cuda, compute_dtype=torch.float32)
    assert m.loss_fn.kind == 'dice' and m.model.loss == 'dice'
    m.train()
    out = m.training_step((img.to(cuda), mask.to(cuda)))
    with torch.no_grad():
        z = m.model(img.to(cuda), normalize=True, mean=MEAN, std=STD).cpu()   # (train mode: same batch statistics)
    assert abs(out['loss'].item() - DiceLoss()(z, mask).item()) <= 1e-5
    m.model.loss = 'bce'                       # same plan object, new criterion
    out = m.training_step((img.to(cuda), mask.to(cuda)))
    out['loss'].backward()
    assert abs(out['loss'].item() - bce_with_logits(z, mask).item()) <= 2e-5
    assert torch.isfinite(m.model.arena.grad).all() and float(m.model.arena.grad.abs().max()) > 0
    with pytest.raises(ValueError):
        OCTSegmentationModel('Unet', 'resnet18', 'u', 3, ['Lumen'], device=cuda, loss='focal')


def test_bce_bf16_704_properties(cuda):
    """BASELINE frame size, bf16: the BCE value recomputed in float64 from the engine's own logits, gradient linear in grad_scale."""
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unetplusplus', 'resnet34', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=3, loss='dice+bce')
    net.train()
    img, mask = make_batch(2, 1, 704, seed=8)
    img, mask = img.to(cuda), mask.to(cuda)
    loss, logits, _ = net.train_step_raw(img, mask, normalize=True, mean=MEAN, std=STD)
    g1 = net.arena.grad.clone()
    from oracle import DiceBCELoss
    want = DiceBCELoss()(logits.double().cpu(), mask.double().cpu()).item()
    assert abs(loss.item() - want) <= 2e-6 * max(1.0, want)
    net.load_state_dict(net.state_dict())      # same weights; running statistics moved, which a train-mode step does not read
    loss2, _, _ = net.train_step_raw(img, mask, normalize=True, mean=MEAN, std=STD, grad_scale=0.5)
    g2 = net.arena.grad
    cos = torch.nn.functional.cosine_similarity(g1.double().flatten(), g2.double().flatten(), dim=0).item()
    ratio = (g2.double().norm() / g1.double().norm()).item()
    print(f'bf16 704: loss {loss.item():.6f} (float64 {want:.6f}), grad_scale 0.5: cosine {cos:.6f} norm ratio {ratio:.4f}')
    assert cos > 0.999 and abs(ratio - 0.5) < 0.01


def test_noncontiguous_frame_survives_until_backward(cuda):
    """with the thin.hip stem (bf16) the plan keeps the raw frame pointer of the training forward and the backward reads
    it again; for a channels_last / permuted input that pointer is the contiguous COPY made on entry, which must stay alive (and
    unrecycled) until loss.backward().  Allocations between forward and backward would otherwise overwrite it."""
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=1)
    net.train()
    img, mask = make_batch(4, 1, 128, seed=9)
    img, mask = img.to(cuda), mask.to(cuda)
    loss, _, _ = net.dice_step(img, mask, normalize=True, mean=MEAN, std=STD)
    loss.backward()
    g_ref = net.arena.grad.clone()
    net.arena.grad = None
    nhwc = img.permute(0, 2, 3, 1).contiguous()           # the reference's dataset yields HWC frames; a view back to NCHW is not contiguous
    view = nhwc.permute(0, 3, 1, 2)
    assert not view.is_contiguous()
    loss2, _, _ = net.dice_step(view, mask, normalize=True, mean=MEAN, std=STD)
    junk = [torch.full_like(img, float(i)) for i in range(6)]      # recycle any freed block of the frame's size
    torch.cuda.synchronize()
    loss2.backward()
    del junk
    w = next(p for p in net.param_table if p['name'] == 'encoder.conv1.weight')
    a = g_ref[w['offset']:w['offset'] + w['numel']].double()
    b = net.arena.grad[w['offset']:w['offset'] + w['numel']].double()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    print(f'stem weight gradient, contiguous vs permuted frame: cosine {cos:.8f}')
    assert abs(loss.item() - loss2.item()) < 1e-6 and cos > 0.9999     # (bf16 atomics: not bit-identical)
