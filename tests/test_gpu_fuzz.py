"""Random-shape train-step parity of the engine against the CPU oracle (the case generator of
tools/fuzz_parity.py, now part of `pytest -m gpu`).

fp32: 40 seeded cases (arch x encoder x batch x classes x non-square frame), kink-free BN biases so that
gradients are a smooth function of the rounding.  A case passes outright when it meets the tight bounds
against the fp32 oracle.  When a gradient does not, the oracle is re-run in float64 on the same weights and
inputs and the engine is judged against THAT: two fp32 implementations with different summation orders are
both a rounding distance away from the exact gradient, and a saturated net (|logits| > 17, Dice gradient
~ exp(-|z|)) amplifies that distance for both.  The engine's error against float64 may not exceed a few
times the fp32 oracle's own error against float64.

Round 1's three out-of-bounds cases of this very sequence (k = 17, 19, 31; gpurun_out/fuzz3.log) are in it:
k = 19 (unet/resnet18 B=3 C=1 160x64, cosine 0.99899) was a defect -- dp/dz written as p*(1-p) vanished
for z > 17; the other two are rounding of 50-layer encoders (float64 triage below).

bf16: 12 cases incl. resnet50 against the fp32 oracle with the loose bounds bf16 storage allows.
"""
import numpy as np
import pytest
import torch

from synth import make_batch
from test_gpu_net import _grad_report

pytestmark = pytest.mark.gpu

FUZZ_SEED = 21          # the sequence of round 1's fuzz3.log
N_FP32, N_BF16 = 40, 12


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        arch = ['unet', 'unetplusplus', 'linknet'][rng.integers(3)]
        enc = ['resnet18', 'resnet34', 'resnet50'][rng.integers(3)]
        B = int(rng.integers(2, 5)); classes = int(rng.integers(1, 5))
        H, W = 32 * int(rng.integers(2, 8)), 32 * int(rng.integers(2, 8))
        out.append((k, arch, enc, B, classes, H, W))
    return out


def _build(k, arch, enc, classes, residual_safe=False):
    """BN biases at +-8: a pre-activation relu(bn(y)) is then >= ~4 sigma away from its kink.  That does NOT cover the residual
    adds relu(bn(y) + identity) -- the identity is an unbounded non-negative tensor, so bias -8 + identity ~ 8 lands on the kink.
    residual_safe=True gives every block-final BatchNorm (BasicBlock bn2, Bottleneck bn3, downsample.1) bias +8: block outputs
    are then always-on and the whole net is smooth in the rounding."""
    from oracle import create_model
    from oracle.nets import randomize_bn
    torch.manual_seed(100 + k)
    ref = create_model(arch, enc, classes=classes)
    randomize_bn(ref, 100 + k)
    g = torch.Generator().manual_seed(200 + k)
    last = 'bn3' if enc in ('resnet50', 'resnet101', 'resnet152') else 'bn2'
    with torch.no_grad():
        for name, m in ref.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.bias.copy_(8.0 * ((torch.rand(m.bias.shape, generator=g) < 0.7).float() * 2 - 1))
                if residual_safe and name.startswith('encoder.layer') and (name.endswith('.' + last) or name.endswith('downsample.1')):
                    m.bias.fill_(8.0)
    return ref.train()


def _near_kinks(ref64, img64, reach=5e-5):
    """Number of ReLU inputs of the float64 oracle within fp32-rounding reach of 0 (terms of magnitude ~16 summed over ~50 layers)."""
    hits = []
    hooks = [m.register_forward_pre_hook(lambda mod, inp: hits.append(int((inp[0].detach().abs() < reach).sum())))
             for m in ref64.modules() if isinstance(m, torch.nn.ReLU)]
    z = ref64(img64)
    for h in hooks:
        h.remove()
    return sum(hits), z


def _inputs(k, B, classes, H, W):
    img, mask = make_batch(B, classes, max(H, W), seed=300 + k)
    return img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()


def _worst_vs(grads, truth, gmax):
    """max over parameters of max|g - truth| / max(max|truth|, 1e-3 * global max), and the squared-error sums of the cosine."""
    worst, name = 0.0, ''
    for n, t in truth.items():
        e = (grads[n].double() - t).abs().max().item() / max(t.abs().max().item(), 1e-3 * gmax)
        if e > worst:
            worst, name = e, n
    return worst, name


def _cos(a, b):
    num = sum(float((a[n].double() * b[n]).sum()) for n in b)
    da = sum(float((a[n].double() ** 2).sum()) for n in b)
    db = sum(float((b[n] ** 2).sum()) for n in b)
    return num / (da ** 0.5 * db ** 0.5 + 1e-300)


_FP32 = _cases(N_FP32, FUZZ_SEED)
# a second seeded sequence (round 3: the suite has the time since the oracle's thread count follows the cgroup quota); k = 100 ..
_FP32 += [(100 + c[0],) + c[1:] for c in _cases(24, FUZZ_SEED + 100)]


@pytest.mark.parametrize('case', _FP32, ids=[f'k{c[0]}-{c[1]}-{c[2]}-B{c[3]}-C{c[4]}-{c[5]}x{c[6]}' for c in _FP32])
def test_fuzz_train_step_fp32(cuda, case):
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    k, arch, enc, B, classes, H, W = case
    ref = _build(k, arch, enc, classes)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32)
    net.load_state_dict(ref.state_dict())
    net.train()
    img, mask = _inputs(k, B, classes, H, W)
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    err = (logits.cpu() - z.detach()).abs().max().item()
    scale = z.detach().abs().max().item()
    grads = {n: g.cpu() for n, g in net.named_grads().items()}
    cos, worst, name = _grad_report(grads, ref)
    print(f'k={k} {arch}/{enc} B={B} C={classes} {H}x{W}: logits {err:.1e}/{scale:.1f} loss {abs(loss.item() - loss_ref.item()):.1e} '
          f'cos {cos:.8f} worst {worst:.1e} ({name})')
    assert err <= 2e-4 * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    if cos > 0.999999 and worst < 5e-3:
        return
    # ---- triage against the exact gradient (float64 oracle, same weights / inputs)
    ref64 = _build(k, arch, enc, classes).double()
    near, z64 = _near_kinks(ref64, img.double())
    DiceLoss()(z64, mask.double()).backward()
    truth = {n: p.grad for n, p in ref64.named_parameters()}
    gmax = max(t.abs().max().item() for t in truth.values())
    o32 = {n: p.grad for n, p in ref.named_parameters()}
    w_eng, n_eng = _worst_vs(grads, truth, gmax)
    w_o32, n_o32 = _worst_vs(o32, truth, gmax)
    c_eng, c_o32 = _cos(grads, truth), _cos(o32, truth)
    print(f'   vs float64: engine worst {w_eng:.2e} ({n_eng}) cos {c_eng:.9f} | fp32 oracle worst {w_o32:.2e} ({n_o32}) cos {c_o32:.9f} | '
          f'{near} ReLU inputs within 5e-5 of the kink')
    if near == 0:   # smooth point: the engine may be a few times further from the exact gradient than the fp32 oracle, no more
        assert w_eng <= max(5e-3, 4.0 * w_o32), f'engine gradient {n_eng} is {w_eng:.2e} from float64, the fp32 oracle only {w_o32:.2e}'
        assert 1.0 - c_eng <= max(1e-6, 4.0 * (1.0 - c_o32))
    else:
        # a ReLU input (a residual add: see _build) within rounding reach of 0: its mask may legitimately differ between two fp32
        # implementations, and one flipped pixel of a 16 x 8 x B map moves that channel's gradients by ~1 / (128 B).  Bounded, not
        # excused: a flip cannot move the global cosine; test_fuzz_residual_safe_fp32 runs the same shapes without such kinks
        assert w_eng <= 3e-2 and c_eng >= 0.99999, f'{n_eng}: {w_eng:.2e}, cosine {c_eng:.7f} with {near} near-kink inputs'


_SAFE = [c for c in _cases(80, FUZZ_SEED) if c[2] == 'resnet50'][:12]


@pytest.mark.parametrize('case', _SAFE, ids=[f'k{c[0]}-{c[1]}-{c[2]}-B{c[3]}-C{c[4]}-{c[5]}x{c[6]}' for c in _SAFE])
def test_fuzz_residual_safe_fp32(cuda, case):
    """The bottleneck-encoder shapes of the sequence with block-final BN biases at +8 (no ReLU input can reach its kink, residual
    adds included): every parameter gradient within 2e-3 of its largest element, cosine 1 - 1e-6 -- the bound the 18/34-layer
    nets meet.  This is the experiment that separates 'ReLU mask flipped at a residual add' from 'the engine is imprecise on
    bottleneck nets' for the out-of-bound cases of test_fuzz_train_step_fp32 (k = 17, 31 of the sequence)."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    k, arch, enc, B, classes, H, W = case
    ref = _build(k, arch, enc, classes, residual_safe=True)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32)
    net.load_state_dict(ref.state_dict())
    net.train()
    img, mask = _inputs(k, B, classes, H, W)
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    err = (logits.cpu() - z.detach()).abs().max().item()
    scale = z.detach().abs().max().item()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'k={k} residual-safe {arch}/{enc} B={B} C={classes} {H}x{W}: logits {err:.1e}/{scale:.1f} cos {cos:.8f} worst {worst:.1e} ({name})')
    assert err <= 2e-4 * max(1.0, scale) and abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos > 0.999999 and worst < 2e-3


_BF16 = _cases(N_BF16, FUZZ_SEED + 1)
if not any(c[2] == 'resnet50' for c in _BF16):   # the sequence must exercise the bottleneck encoder
    _BF16[-1] = (_BF16[-1][0], 'unetplusplus', 'resnet50') + _BF16[-1][3:]


@pytest.mark.parametrize('case', _BF16, ids=[f'k{c[0]}-{c[1]}-{c[2]}-B{c[3]}-C{c[4]}-{c[5]}x{c[6]}' for c in _BF16])
def test_fuzz_train_step_bf16(cuda, case):
    """bf16 engine against the fp32 oracle: logits within 3 % of their scale (or 1.5 x the error of torch's CPU bf16 autocast of the oracle, whichever is larger), Dice loss within 1e-3 (2.5e-3 for batches of
    fewer than 40 k pixels: few pixels per class), global gradient cosine >= 0.999 (kink-free nets)."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    k, arch, enc, B, classes, H, W = case
    ref = _build(k, arch, enc, classes)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.bfloat16)
    net.load_state_dict(ref.state_dict())
    net.train()
    img, mask = _inputs(k, B, classes, H, W)
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    err = (logits.cpu() - z.detach()).abs().max().item()
    scale = z.detach().abs().max().item()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    # yardstick for the small logit scales LinkNet produces at these sizes: what torch's own CPU bf16 autocast loses on the same net and batch
    # (a 40-case run of tools/fuzz_parity.py with another seed had three LinkNet cases at 3.2 .. 4.4 % -- each BELOW its autocast error)
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        err_ac = (ref(img).float() - z.detach()).abs().max().item()
    print(f'k={k} bf16 {arch}/{enc} B={B} C={classes} {H}x{W}: logits {err:.1e}/{scale:.1f} (autocast {err_ac:.1e}) loss {abs(loss.item() - loss_ref.item()):.1e} cos {cos:.6f}')
    assert err <= max(3e-2 * max(1.0, scale), 1.5 * err_ac)
    # bounds sit at what is measured, not 10-100x above it (two 40-case runs, profiles/r2_fuzz_bf16_seed777.txt: cosine 0.9993 .. 0.999999,
    # Dice loss error <= 1.3e-3, <= 1e-3 from 40 k pixels per batch upwards): a regression that halves bf16 gradient quality turns red
    assert abs(loss.item() - loss_ref.item()) <= (1e-3 if B * H * W >= 40000 else 2.5e-3)
    assert cos > 0.999
