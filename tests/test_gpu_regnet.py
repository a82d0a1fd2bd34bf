"""timm RegNet encoders through smp's RegNetEncoder (reference sweep: configs/tune.yaml:19-24 -> smp.create_model(arch, 'timm-regnetx_002' |
'timm-regnetx_064'), src/models/smp/model.py:38-44).  The oracle restates timm 0.9.2's RegNet (oracle/nets.py, pinned by the published
parameter counts in tests/test_oracle.py); the engine runs the grouped 3x3 convs as per-group launches on channel slices and the 3x3 stem
through im2col rows.  Same bounds as the ResNet nets (tests/test_gpu_net.py): fp32 logits 1e-4 of their scale, Dice 1e-5, counts exact,
running statistics 1e-4, every gradient of the kink-free net within 2e-3 of its largest element, cosine >= 0.999999."""
import pytest
import torch

import test_gpu_net as T
from synth import make_batch

pytestmark = pytest.mark.gpu

# With ReLU kinks present the forward band is wider than resnet18's: the narrow stages (24 .. 368 channels) end in 2 x 2 maps here, whose
# BatchNorms see 8 values per channel -- one flipped ReLU mask moves a whole channel's statistics.  Measured 2.6e-4 / 1.1e-4 of the logit
# scale (regnetx_002) against 8e-7 on the same nets kink-free (cosine 1.00000000, every gradient within 5e-5): flips, not arithmetic.
T.FWD_TOL.setdefault('timm-regnetx_002', 4e-4)
T.FWD_TOL.setdefault('timm-regnetx_064', 2e-4)
T.FWD_TOL.setdefault('timm-regnety_120', 2e-4)

NETS = [('unet', 'timm-regnetx_002', 1, 2, 64), ('unetplusplus', 'timm-regnetx_002', 2, 3, 64), ('unet', 'timm-regnetx_064', 1, 2, 64),
        ('unet', 'timm-regnetx_002', 2, 2, 96),
        # RegNetY: squeeze-excite gates behind every grouped conv (mean -> fc1 -> ReLU -> fc2 -> sigmoid); LinkNet fits it (2240 / 4 = 560)
        ('unet', 'timm-regnety_120', 1, 2, 64), ('linknet', 'timm-regnety_120', 2, 3, 64)]
IDS = ['-'.join(map(str, c)) for c in NETS]


@pytest.mark.parametrize('cfg', NETS, ids=IDS)
def test_regnet_train_step_parity_fp32(cuda, cfg):
    T.test_train_step_parity_fp32(cuda, cfg)


@pytest.mark.parametrize('cfg', NETS, ids=IDS)
def test_regnet_gradients_kinkfree_fp32(cuda, cfg):
    T.test_gradients_kinkfree_fp32(cuda, cfg)


@pytest.mark.parametrize('cfg', NETS[:3] + NETS[4:5], ids=IDS[:3] + IDS[4:5])
def test_regnet_eval_forward_and_keys_fp32(cuda, cfg):
    from oct_segmentation_amd.engine import SegNet
    arch, enc, classes, B, S = cfg
    ref = T._oracle(arch, enc, classes).eval()
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32).eval()
    sd, rd = net.state_dict(), ref.state_dict()
    assert sorted(sd.keys()) == sorted(rd.keys())                          # timm's module tree, key for key
    assert all(tuple(sd[k].shape) == tuple(rd[k].shape) for k in rd), [k for k in rd if tuple(sd[k].shape) != tuple(rd[k].shape)][:4]
    net.load_state_dict(rd)
    back = net.state_dict()
    assert all(torch.equal(back[k].cpu(), rd[k]) for k in rd if rd[k].dtype.is_floating_point)   # grouped weights survive the split / join
    img, _ = make_batch(B, classes, S, seed=5)
    with torch.no_grad():
        y_ref = ref(img)
    y = net(img.to(cuda), normalize=False).cpu()
    scale = y_ref.abs().max().item()
    err = (y - y_ref).abs().max().item()
    print(f'{cfg} eval: logits {err:.2e} / {scale:.2f}')
    assert err <= 1e-4 * max(1.0, scale)


def test_regnet_larger_frame_fp32(cuda):
    """256 x 320: several tiles per map, every stage's grouped conv on maps from 128 x 160 down to 8 x 10 (stride-2 first blocks included)."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    ref = T._oracle('unet', 'timm-regnetx_002', 2, seed=3, kinkfree=True).train()
    net = SegNet('unet', 'timm-regnetx_002', classes=2, device=cuda, compute_dtype=torch.float32).train()
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(2, 2, 320, seed=5)
    img, mask = img[:, :, :256].contiguous(), mask[:, :, :256].contiguous()
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    err = (logits.cpu() - z.detach()).abs().max().item()
    cos, worst, name = T._grad_report(net.named_grads(), ref)
    print(f'unet/regnetx_002 256x320: logits {err:.2e} / {z.abs().max().item():.1f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, z.abs().max().item()) and abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos >= 0.999999
    if worst >= 2e-3:      # (measured 2.6e-3 on encoder.s4.b1.conv3.conv.weight, an 8 x 10 map: re-judged against a float64 run of the oracle)
        from test_gpu_deeplab import judge_gradients
        judge_gradients(ref, net.named_grads(), img, mask, tag='unet/regnetx_002 256x320: ', normalize=False, max_rejudged=4)


@pytest.mark.parametrize('cfg', [('unet', 'timm-regnetx_002', 1, 2, 256), ('unetplusplus', 'timm-regnetx_064', 1, 2, 256), ('unet', 'timm-regnety_120', 1, 2, 256)],
                         ids=lambda c: '-'.join(map(str, c)))
def test_regnet_bf16_engine_vs_fp32_oracle(cuda, cfg):
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    arch, enc, classes, B, S = cfg
    ref = T._oracle(arch, enc, classes, seed=13, kinkfree=True).train()
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.bfloat16).train()
    net.load_state_dict(ref.state_dict())
    img, mask = make_batch(B, classes, S, seed=17)
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    scale = z.detach().abs().max().item()
    err = (logits.cpu() - z.detach()).abs().max().item()
    cos, worst, name = T._grad_report(net.named_grads(), ref)
    print(f'{cfg} bf16: logits {err:.2e}/{scale:.1f} ({err / max(scale, 1):.2%}), Dice loss {loss.item():.6f} vs {loss_ref.item():.6f}, grad cosine {cos:.5f}')
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 and cos >= 0.999 and err <= 3e-2 * max(1.0, scale)


def test_regnet_704_bf16_properties_and_unsupported_pairs(cuda):
    """BASELINE frame size in bf16: finite, Dice recomputed in float64 from the engine's own logits; the pairs the engine refuses say why."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unetplusplus', 'timm-regnetx_002', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=2).train()
    img, mask = (t.to(cuda) for t in make_batch(2, 1, 704, seed=4))
    loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    assert torch.isfinite(logits).all() and torch.isfinite(net.arena.grad).all()
    want = DiceLoss()(logits.double().cpu(), mask.double().cpu()).item()
    assert abs(loss.item() - want) <= 2e-6
    assert int(stats.sum()) == 2 * 704 * 704
    with pytest.raises((KeyError, RuntimeError)):
        SegNet('linknet', 'timm-regnetx_002', classes=1, device=cuda)        # decoder widths 368 / 4 = 92: not a multiple of the 8-channel vector
    with pytest.raises((KeyError, RuntimeError)):
        SegNet('deeplabv3plus', 'timm-regnetx_002', classes=1, device=cuda)  # dilated RegNet stages: not built
    with pytest.raises((KeyError, RuntimeError)):
        SegNet('pspnet', 'timm-regnetx_064', classes=1, device=cuda)         # pyramid branches on 392 / 4 = 98 channels


@pytest.mark.parametrize('arch', ['fpn'])
def test_regnet_under_the_sweep_decoders_fp32(cuda, arch):
    """FPN (all five features, lateral 1x1 convs on 24 .. 368 channels) over timm-regnetx_002, kink-free, dropout pattern injected.
    (PSPNet and LinkNet run branches on a quarter of a feature's channels -- 14, 92 -- which the 8-channel NHWC vectors cannot hold: refused.)"""
    import test_gpu_fpn, test_gpu_pspnet
    pair = test_gpu_fpn._pair if arch == 'fpn' else test_gpu_pspnet._pair
    ref, net, img, mask, z, loss_ref, logits, loss, stats = pair(cuda, 'timm-regnetx_002', 2, 3, 64, 96, seed=7, kinkfree=True)
    grads = net.named_grads()
    cos, worst, name = test_gpu_pspnet._report(grads, ref) if arch == 'pspnet' else T._grad_report(grads, ref)
    err = (logits - z).abs().max().item()
    print(f'{arch}/timm-regnetx_002: logits {err:.2e} / {z.abs().max().item():.2f}, grad cosine {cos:.9f}, worst {worst:.2e} ({name})')
    assert err <= 1e-4 * max(1.0, z.abs().max().item()) and abs(loss.item() - loss_ref.item()) <= 1e-5
    assert cos >= 0.999999 and worst < 2e-3
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())
