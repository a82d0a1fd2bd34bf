"""Every BASELINE.json configuration through `pytest -m gpu`.

  c1  U-Net/resnet18 1-class 256x256 fp32 B=2            -> tests/golden/c1_unet_resnet18_256.npz (test_gpu_golden.py)
  c2  U-Net++/resnet101 1-class 704x704 bf16 B=16        -> full-size properties here + golden unetplusplus_resnet101 (fp32, 64x64)
  c3  LinkNet/resnet50 2-class 704x704 bf16 16 per GPU   -> full-size properties here
  c4  U-Net/resnet50 1-class 704x704 bf16 + on-GPU augmentation -> full-size properties here (through octseg_augment)
  c5  LM + FC_LC + VV ensemble, 704x704, replayed hipGraph -> segment() replay == eager here

plus the bf16 engine -- the benchmarked dtype -- against the fp32 oracle at >= 256x256 on the bottleneck encoder
(U-Net++/resnet50, LinkNet/resnet50): Dice loss <= 1e-3 (north_star), logits <= 3 % of their scale, global gradient
cosine >= 0.9995 (kink-free BN biases: ReLU masks are then stable under bf16 rounding).

Full-size cases cannot run the CPU oracle in reasonable time (45 TFLOP per step), so they check size-independent
properties: finite outputs, a deterministic forward, eval-mode batch-permutation equivariance, confusion counts equal
to a recount from the returned logits, tp+fp+fn+tn = pixels, Dice loss recomputed in float64 from the returned logits,
gradient linear in grad_scale.
"""
import json
import os

import numpy as np
import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu


def _dice_f64(logits, mask):
    """smp DiceLoss(multilabel, from_logits) in float64 from the engine's own logits."""
    p = torch.sigmoid(logits.double())
    t = mask.double()
    inter = (p * t).sum(dim=(0, 2, 3)); card = (p + t).sum(dim=(0, 2, 3))
    loss = (1.0 - 2.0 * inter / card.clamp_min(1e-7)) * (t.sum(dim=(0, 2, 3)) > 0)
    return loss.mean().item()


def _full_size_properties(cuda, arch, enc, classes, B, augment=False, S=704):
    from oct_segmentation_amd.engine import SegNet
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.bfloat16, seed=3)
    img, mask = make_batch(B, classes, S, seed=21, empty_last=(classes > 1))
    img, mask = img.to(cuda), mask.to(cuda)
    if augment:   # cfg #4: the batch goes through the reference's eight transforms on the GPU first
        from oct_segmentation_amd import augment as A
        params = A.sample_params(B, S, np.random.default_rng(7))
        img, mask = A.augment(img, mask, params)
        assert img.shape == (B, 3, S, S) and mask.shape == (B, classes, S, S)
        assert torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 255.0
        assert bool(((mask == 0) | (mask == 1)).all())
        assert torch.equal(img, img.round())                       # the uint8 grid of the reference's images
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    net.eval()
    y1 = net(img, normalize=True, mean=mean, std=std)
    y2 = net(img, normalize=True, mean=mean, std=std)
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(cuda)
    assert torch.equal(net(img[perm], normalize=True, mean=mean, std=std), y1[perm])
    del y2
    net.train()
    loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=mean, std=std, grad_scale=1.0)
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all() and torch.isfinite(net.arena.grad).all() and np.isfinite(loss.item())
    g1 = net.arena.grad.clone()
    s = stats.cpu()
    pred = logits.sigmoid() > 0.5
    t = mask > 0
    assert torch.equal(s[..., 0], (pred & t).sum(dim=(2, 3)).cpu())
    assert torch.equal(s[..., 1], (pred & ~t).sum(dim=(2, 3)).cpu())
    assert torch.equal(s[..., 2], (~pred & t).sum(dim=(2, 3)).cpu())
    assert torch.equal(s.sum(-1), torch.full_like(s[..., 0], S * S))
    assert abs(loss.item() - _dice_f64(logits, mask)) <= 2e-6      # fp32 sigmoid, f64 sums in the kernel
    assert 0.0 <= loss.item() <= 1.0
    nz = float((g1 != 0).float().mean())
    assert nz > 0.99, f'only {nz:.2f} of the gradient arena is non-zero'
    buf = net.bn_buffers.clone()
    net.bn_buffers.zero_()
    loss2, logits2, _ = net.train_step_raw(img, mask, normalize=True, mean=mean, std=std, grad_scale=0.5)
    assert torch.equal(logits2, logits) and abs(loss2.item() - loss.item()) < 1e-6   # train forward is deterministic too
    ratio = (net.arena.grad.norm() / g1.norm()).item()
    cosg = torch.nn.functional.cosine_similarity(net.arena.grad.flatten(), g1.flatten(), dim=0).item()
    assert abs(ratio - 0.5) < 2e-2 and cosg > 0.995, (ratio, cosg)   # bf16 dL/dlogits rounding + atomics order
    assert not torch.equal(buf, torch.zeros_like(buf))
    return net


def test_c2_unetplusplus_resnet101_704_bf16_b16(cuda):
    """BASELINE configs[1] -- the benchmarked configuration itself."""
    net = _full_size_properties(cuda, 'unetplusplus', 'resnet101', 1, 16)
    assert abs(net.fwd_macs(16, 704, 704) / 16 / 1e9 - 471.07) < 0.01
    del net
    torch.cuda.empty_cache()


def test_c3_linknet_resnet50_2class_704_bf16_b16(cuda):
    """BASELINE configs[2]: one rank's share (16 frames) of the 128-frame data-parallel batch."""
    net = _full_size_properties(cuda, 'linknet', 'resnet50', 2, 16)
    assert abs(net.fwd_macs(16, 704, 704) / 16 / 1e9 - 54.99) < 0.01
    del net
    torch.cuda.empty_cache()


def test_c4_unet_resnet50_704_bf16_with_gpu_augmentation(cuda):
    """BASELINE configs[3]: U-Net 1-class with the heavy augmentation on the GPU in front of the step."""
    net = _full_size_properties(cuda, 'unet', 'resnet50', 1, 16, augment=True)
    assert abs(net.fwd_macs(16, 704, 704) / 16 / 1e9 - 80.41) < 0.01
    del net
    torch.cuda.empty_cache()


def test_c5_ensemble_704_graph_replay(cuda, tmp_path):
    """BASELINE configs[4]: LM (U-Net++/resnet101) + FC_LC (LinkNet/resnet50, 2 classes) + VV (U-Net/resnet50) on 704x704 frames in
    fp16 through segment() with every eval forward replayed from a captured hipGraph: equal to the eager masks bit for bit, the
    4-channel stack follows CLASS_IDS - 1 and the FC_LC channel mapping of MODELS_META (predict.py:23-28)."""
    from PIL import Image
    from oct_segmentation_amd.model import OCTSegmentationModel
    from oct_segmentation_amd.predict import segment
    specs = {'LM': ('unetplusplus', 'resnet101', ['Lumen']), 'FC_LC': ('linknet', 'resnet50', ['Lipid core', 'Fibrous cap']),
             'VV': ('unet', 'resnet50', ['Vasa vasorum'])}
    for i, (d, (arch, enc, classes)) in enumerate(specs.items()):
        os.makedirs(os.path.join(tmp_path, d))
        m = OCTSegmentationModel(arch, enc, f'{arch}_{enc}', 3, classes, device=cuda, seed=40 + i, compute_dtype=torch.bfloat16)
        m.save_checkpoint(os.path.join(tmp_path, d, 'weights.ckpt'))
        with open(os.path.join(tmp_path, d, 'config.json'), 'w') as f:
            json.dump({'model_name': f'{arch}_{enc}', 'architecture': arch, 'encoder': enc, 'input_size': 704, 'classes': classes}, f)
        del m
    torch.cuda.empty_cache()
    rng = np.random.default_rng(0)
    base = make_batch(3, 1, 750, seed=9)[0].permute(0, 2, 3, 1).numpy().astype(np.uint8)[:, :, :, ::-1]   # RGB 750x750 frames
    images = [Image.fromarray(np.ascontiguousarray(b)) for b in base]
    classes = ['Lumen', 'Fibrous cap', 'Lipid core', 'Vasa vasorum']
    outs = []
    for use_graph in (False, True):
        masks = [np.zeros((1000, 1000, 4)) for _ in images]
        outs.append(segment(images, masks, [1000, 1000], classes, str(tmp_path), device='cuda', batch_size=1, use_graph=use_graph,
                            compute_dtype=torch.float16))
    for a, b in zip(*outs):
        assert a.shape == (1000, 1000, 4) and set(np.unique(a)) <= {0.0, 1.0}
        assert np.array_equal(a, b)
    frac = np.mean([o.mean() for o in outs[0]])
    assert 0.0 < frac < 1.0
    _ = rng


BF16_PARITY = [('unetplusplus', 'resnet50', 1, 2, 256), ('linknet', 'resnet50', 2, 2, 256), ('unetplusplus', 'resnet101', 1, 2, 256),
               # 9 x 64 x 64 = 288 M tiles in layer1: the persistent 1x1 GEMM (gemm1x1.hip) walks several tiles per workgroup with
               # its BatchNorm partial sums carried across them (conv3: two N tiles on 256 workgroup rows)
               ('unet', 'resnet50', 1, 9, 256),
               # the benchmark's network at the benchmark's frame size (two frames): bf16 engine against the fp32 oracle
               ('unetplusplus', 'resnet101', 1, 2, 704)]


@pytest.mark.parametrize('cfg', BF16_PARITY, ids=['-'.join(map(str, c)) for c in BF16_PARITY])
def test_bf16_engine_vs_fp32_oracle_bottleneck_256(cuda, cfg):
    from oracle import DiceLoss, get_stats
    from oct_segmentation_amd.engine import SegNet
    from test_gpu_net import _oracle, _grad_report
    arch, enc, classes, B, S = cfg
    ref = _oracle(arch, enc, classes, seed=13, kinkfree=True).train()
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.bfloat16)
    net.load_state_dict(ref.state_dict())
    net.train()
    img, mask = make_batch(B, classes, S, seed=17)
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    # yardstick: torch's own CPU bf16 autocast of the same network (bf16 conv inputs, fp32 BatchNorm)
    ac = _oracle(arch, enc, classes, seed=13, kinkfree=True).train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        z_ac = ac(img)
    z_ac = z_ac.float().detach()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    scale = z.detach().abs().max().item()
    err = (logits.cpu() - z.detach()).abs().max().item()
    err_ac = (z_ac - z.detach()).abs().max().item()
    cos, worst, name = _grad_report(net.named_grads(), ref)

    def hard_dice(lg):
        tp, fp, fn, tn = get_stats((lg.sigmoid() > 0.5).long(), mask.long())
        return (2 * tp.sum().item()) / max(1, (2 * tp + fp + fn).sum().item())
    d_ref, d_eng, d_ac = hard_dice(z.detach()), hard_dice(logits.cpu()), hard_dice(z_ac)
    print(f'{cfg}: bf16 logits {err:.2e}/{scale:.1f} ({err / max(scale, 1):.2%}; torch autocast {err_ac:.2e}), Dice loss {loss.item():.6f} vs '
          f'{loss_ref.item():.6f}, hard Dice {d_eng:.5f} vs {d_ref:.5f} (autocast {d_ac:.5f}), grad cosine {cos:.5f}')
    assert abs(loss.item() - loss_ref.item()) <= 1e-3            # north_star: Dice within 1e-3 of the reference
    assert cos >= 0.9995                                         # (measured 0.99994 .. 0.999999 on these nets)
    # element-wise logits and the thresholded masks of a randomly initialised net (hard Dice ~0.3: most pixels sit near the
    # threshold): 3 % of the logit scale / 1e-3, or what torch's own bf16 autocast of the same net deviates by
    assert err <= max(3e-2 * max(1.0, scale), 1.5 * err_ac)
    assert abs(d_eng - d_ref) <= max(1e-3, 1.5 * abs(d_ac - d_ref))


@pytest.mark.parametrize('cfg', [('unet', 'resnet50', 1, 4, 128), ('linknet', 'resnet50', 2, 4, 128), ('fpn', 'resnet50', 1, 4, 128),
                                 # 352^2 frames: maps of 88, 44, 22 and 11 pixels -- multiples of 11 and not of 16: the 3x3 layers of the
                                 # encoder's last three stages and of the deep decoder blocks run on 11 x 11 pixel tiles (conv_mfma.hip LOOP_T11)
                                 ('unet', 'resnet50', 1, 2, 352)],
                         ids=lambda c: '-'.join(map(str, c)))
def test_bf16_every_weight_gradient_direction(cuda, cfg):
    """Per-PARAMETER check of the bf16 engine against the fp32 oracle (kink-free nets): the cosine of every conv / ConvTranspose weight
    gradient.  The global cosine of test_bf16_engine_vs_fp32_oracle_bottleneck_256 is dominated by the decoder; this one sees a single
    wrong layer -- in particular the bottleneck conv3 weight gradients, whose x operand gets its lazy BatchNorm + ReLU applied to the
    MFMA fragment inside the LDS-DMA ring kernel (wgrad1x1.hip, the AFF instantiation; layer1 / layer2 maps have >= 1024 pixels here)."""
    from oracle import DiceLoss
    from oct_segmentation_amd.engine import SegNet
    from test_gpu_net import _oracle
    arch, enc, classes, B, S = cfg
    if arch == 'fpn':
        import test_gpu_fpn
        ref, net, img, mask, z, loss_ref, logits, loss, stats = test_gpu_fpn._pair(cuda, enc, classes, B, S, S, seed=13, kinkfree=True, dtype=torch.bfloat16)
    else:
        ref = _oracle(arch, enc, classes, seed=13, kinkfree=True).train()
        net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.bfloat16)
        net.load_state_dict(ref.state_dict())
        net.train()
        img, mask = make_batch(B, classes, S, seed=17)
        DiceLoss()(ref(img), mask).backward()
        net.train_step_raw(img.to(cuda), mask.to(cuda))
        torch.cuda.synchronize()
    grads = net.named_grads()
    worst, worst_name, n = 1.0, '', 0
    for name, p in ref.named_parameters():
        if p.dim() != 4 or p.grad is None:
            continue
        a, b = grads[name].cpu().double().flatten(), p.grad.double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        n += 1
        if cos < worst:
            worst, worst_name = cos, name
    print(f'{cfg}: {n} conv weights, worst per-parameter gradient cosine {worst:.5f} ({worst_name})')
    assert worst >= 0.95, (worst, worst_name)        # (bf16 noise on the 8 x 8 maps of layer3 reaches 0.986; a mis-mapped operand gives ~0)
