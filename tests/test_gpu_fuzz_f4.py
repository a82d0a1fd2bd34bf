"""Seeded fuzz over the sweep architectures added in round 3 (FPN, DeepLabV3+, PSPNet, DeepLabV3): random encoder x batch x classes x
non-square frame, kink-free normalisation biases (so that fp32 implementations agree on every ReLU mask), the oracle's dropout pattern
injected.  Bounds of the per-architecture tests: logits 1e-4 of their scale (2e-4 behind the 50+-layer encoders), Dice 1e-5, counts exact,
gradient cosine 1 - 1e-6, every parameter within 2e-3 of its largest element or re-judged against float64 (test_gpu_deeplab.judge_gradients).
The existing 40-case fuzz (test_gpu_fuzz.py) keeps its three architectures so that its seeded sequence stays the one of round 1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases(n=18, seed=2024, keep=tuple(range(18))):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        arch = ['fpn', 'deeplabv3plus'][k % 2] if k < 10 else ('pspnet' if k < 15 else 'deeplabv3')   # (appended: earlier draws unchanged)
        enc = ['resnet18', 'resnet34', 'resnet50'][rng.integers(3)]      # (resnet101 cost 35 s per float64 re-judge; its blocks are resnet50's)
        B = int(rng.integers(2, 5)) if arch == 'fpn' else int(rng.integers(3, 5))
        classes = int(rng.integers(1, 4))
        H, W = (int(32 * rng.integers(2, 5)) for _ in range(2))
        if arch == 'deeplabv3':      # output stride 8 with layer4 on 16 nested sub-grids: frames below 96 px leave 2x2-pixel maps
            H, W = max(H, 96), max(W, 96)
        out.append((k, arch, enc, B, classes, H, W))
    return [c for c in out if c[0] in keep]


@pytest.mark.parametrize('case', _cases(), ids=lambda c: f'{c[0]}-{c[1]}-{c[2]}-B{c[3]}-C{c[4]}-{c[5]}x{c[6]}')
def test_fuzz_f4_fp32(cuda, case):
    from oracle import get_stats
    from test_gpu_net import _grad_report
    import test_gpu_deeplab, test_gpu_deeplabv3, test_gpu_fpn, test_gpu_pspnet
    k, arch, enc, B, classes, H, W = case
    pair = {'fpn': test_gpu_fpn._pair, 'deeplabv3plus': test_gpu_deeplab._pair, 'pspnet': test_gpu_pspnet._pair, 'deeplabv3': test_gpu_deeplabv3._pair}[arch]
    ref, net, img, mask, z, loss_ref, logits, loss, stats = pair(cuda, enc, classes, B, H, W, seed=100 + k, kinkfree=True)
    scale = z.abs().max().item()
    err = (logits - z).abs().max().item()
    grads = net.named_grads()
    cos, worst, name = test_gpu_pspnet._report(grads, ref) if arch == 'pspnet' else _grad_report(grads, ref)
    print(f'k={k} {arch}/{enc} B={B} C={classes} {H}x{W}: logits {err:.1e}/{scale:.2f} loss {abs(loss.item() - loss_ref.item()):.1e} cos {cos:.9f} worst {worst:.1e} ({name})')
    assert err <= (2e-4 if enc in ('resnet50', 'resnet101') else 1e-4) * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    tp, fp, fn, tn = get_stats((logits.sigmoid() > 0.5).long(), mask.long())
    away = z.abs() > 1e-3
    if bool(away.all()):
        assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    assert cos >= 0.999999
    if arch == 'fpn':     # (FPN's own frames: test_gpu_fpn._pair does not rescale them, judge_gradients needs the frame the oracle saw)
        assert worst < 2e-3 or test_gpu_deeplab.judge_gradients(ref, grads, img, mask, tag=f'k={k} ') <= 2
    else:
        test_gpu_deeplab.judge_gradients(ref, grads, img, mask, tag=f'k={k} ', max_rejudged=8)


@pytest.mark.parametrize('arch,enc,B,classes,S', [('fpn', 'resnet50', 2, 1, 256), ('deeplabv3plus', 'resnet50', 3, 2, 256), ('pspnet', 'resnet50', 2, 1, 256),
                                                   ('deeplabv3', 'resnet50', 3, 1, 256), ('fpn', 'resnet18', 2, 2, 320)],
                         ids=lambda v: str(v))
def test_f4_bf16_engine_vs_fp32_oracle(cuda, arch, enc, B, classes, S):
    """the sweep architectures in bf16 had property tests and one Dice check only.  Same criterion as the BASELINE nets
    (test_gpu_configs.test_bf16_engine_vs_fp32_oracle_bottleneck_256): bf16 engine against the fp32 oracle on kink-free nets at >= 256^2 --
    Dice within 1e-3 (north_star), gradient cosine >= 0.9995, logits within 2 % of their scale."""
    from test_gpu_net import _grad_report
    import test_gpu_deeplab, test_gpu_deeplabv3, test_gpu_fpn, test_gpu_pspnet
    pair = {'fpn': test_gpu_fpn._pair, 'deeplabv3plus': test_gpu_deeplab._pair, 'pspnet': test_gpu_pspnet._pair, 'deeplabv3': test_gpu_deeplabv3._pair}[arch]
    ref, net, img, mask, z, loss_ref, logits, loss, stats = pair(cuda, enc, classes, B, S, S, seed=41, kinkfree=True, dtype=torch.bfloat16)
    scale = z.abs().max().item()
    err = (logits - z).abs().max().item()
    grads = net.named_grads()
    cos, worst, name = test_gpu_pspnet._report(grads, ref) if arch == 'pspnet' else _grad_report(grads, ref)
    print(f'{arch}/{enc} bf16 {S}^2: logits {err:.2e}/{scale:.2f} ({err / max(scale, 1):.2%}), Dice loss {loss.item():.6f} vs {loss_ref.item():.6f}, grad cosine {cos:.6f}')
    assert abs(loss.item() - loss_ref.item()) <= 1e-3
    assert cos >= 0.9995                           # (measured 0.99964 .. 0.999999)
    assert err <= 2e-2 * max(1.0, scale)           # (measured 0.35 % .. 1.2 % of the scale)
