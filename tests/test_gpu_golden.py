"""GPU replay of the committed golden vectors (tests/golden/*.npz, produced by the CPU oracle in the
build container) through the fp32 HIP engine: logits, Dice loss, confusion counts and the per-parameter
gradient magnitudes must match without the oracle running any arithmetic on the GPU box (it only
re-derives the seeded weights)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('name', ['unet_resnet18', 'unetplusplus_resnet18', 'linknet_resnet18', 'unet_resnet50', 'unetplusplus_resnet50',
                                  'linknet_resnet50', 'unet_resnet18_96x64', 'c1_unet_resnet18_256', 'unetplusplus_resnet101',
                                  'fpn_resnet18_64x96', 'deeplabv3plus_resnet18_64x96', 'pspnet_resnet18_96x64', 'deeplabv3_resnet18_64x96'])
def test_engine_reproduces_golden_vectors(cuda, name):
    from golden.make_golden import CASES, build, case_batch, case_keep, summarize_logits
    from oct_segmentation_amd.engine import SegNet
    arch, enc, classes, B, S, seed = CASES[name]
    g = np.load(os.path.join(GOLDEN, f'{name}.npz'))
    ref = build(arch, enc, classes, seed)           # seeded weights only; no forward on the CPU here
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32)
    net.load_state_dict(ref.state_dict())
    net.train()
    img, mask = case_batch(B, classes, S, seed, arch)
    net.dropout_keep = case_keep(arch, B, S, seed)     # (None for the architectures without dropout)
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda), normalize=True,
                                             mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    torch.cuda.synchronize()
    got = summarize_logits(logits.cpu().numpy())
    if 'logits' in g:
        scale = float(np.abs(g['logits']).max())
        err = float(np.abs(got['logits'] - g['logits']).max())
    else:   # large case: centre crop element-wise, the whole tensor through its float64 sums
        scale = float(g['logits_absmax'])
        err = float(np.abs(got['logits_crop'] - g['logits_crop']).max())
        n = logits.numel()
        assert abs(got['logits_sums'][0] - g['logits_sums'][0]) <= 2e-4 * max(1.0, scale) * n ** 0.5 * 4
        assert abs(got['logits_sums'][2] - g['logits_sums'][2]) <= 2e-4 * max(1.0, scale) * n ** 0.5 * 4
    print(f'{name}: logits max|d| {err:.3e} (scale {scale:.2e}), loss {loss.item():.7f} vs {float(g["loss"]):.7f}')
    assert err <= 2e-4 * max(1.0, scale)
    assert abs(loss.item() - float(g['loss'])) <= 1e-5
    assert np.array_equal(stats.cpu().numpy(), g['stats'])
    grads = net.named_grads()
    sums = np.array([grads[n].abs().sum().item() for n, _ in ref.named_parameters()])
    rel = np.abs(sums - g['grad_abs_sums']) / np.maximum(g['grad_abs_sums'], 1e-3 * g['grad_abs_sums'].max())
    print(f'{name}: worst relative deviation of sum|grad| over parameters {rel.max():.3e}')
    assert rel.max() < 2e-3
