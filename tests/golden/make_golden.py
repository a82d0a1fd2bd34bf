"""Generates the golden vectors under tests/golden from the CPU oracle (run here, in the build
container: `python tests/golden/make_golden.py`).  The reference repository has no importable
implementation of this path in this environment (segmentation_models_pytorch / torchvision / cv2 /
pytorch_lightning are absent), so the vectors pin the ORACLE against silent edits; they are also what
the GPU tests replay through the HIP engine (tests/test_gpu_golden.py).

Each case: seeded weights (torch.manual_seed), seeded synthetic batch, train-mode forward + Dice +
backward.  Stored: logits, loss, per-parameter sum |grad| (in named_parameters order), tp/fp/fn/tn.
Weights are NOT stored (14-32 M floats): they are re-derived from the seed on both sides.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

CASES = {
    # name: (arch, encoder, classes, batch, size, seed)
    'unet_resnet18': ('unet', 'resnet18', 1, 2, 64, 101),
    'unetplusplus_resnet18': ('unetplusplus', 'resnet18', 2, 2, 64, 102),
    'linknet_resnet18': ('linknet', 'resnet18', 2, 2, 64, 103),
    'unet_resnet50': ('unet', 'resnet50', 1, 2, 64, 108),
    # SURVEY.md section 8c list: bottleneck-encoder variants of the other two decoders, a non-square frame, and
    # BASELINE config c1 (U-Net / resnet18, 256 x 256, B = 2) itself
    'unetplusplus_resnet50': ('unetplusplus', 'resnet50', 1, 2, 64, 110),
    'linknet_resnet50': ('linknet', 'resnet50', 2, 2, 64, 111),
    'unet_resnet18_96x64': ('unet', 'resnet18', 2, 2, (96, 64), 112),
    'c1_unet_resnet18_256': ('unet', 'resnet18', 1, 2, 256, 116),
    # the encoder of BASELINE config c2 (U-Net++ / resnet101: 23 bottlenecks in layer3), tiny frame
    'unetplusplus_resnet101': ('unetplusplus', 'resnet101', 1, 2, 64, 120),
    # the sweep architectures of round 3; their dropout keep patterns are derived from the seed (case_keep)
    'fpn_resnet18_64x96': ('fpn', 'resnet18', 2, 2, (64, 96), 140),
    'deeplabv3plus_resnet18_64x96': ('deeplabv3plus', 'resnet18', 2, 4, (64, 96), 141),
    'pspnet_resnet18_96x64': ('pspnet', 'resnet18', 2, 3, (96, 64), 142),
    'deeplabv3_resnet18_64x96': ('deeplabv3', 'resnet18', 1, 4, (64, 96), 143),
}
# Lightning-DDP semantics (reference train.py:122-133 with devices > 1): the global batch is split
# into `world` contiguous shards, every rank runs forward + Dice + backward on ITS shard with local BatchNorm statistics
# and a local Dice loss, the gradients are averaged, BN running buffers follow rank 0.
DDP_CASES = {
    # name: (arch, encoder, classes, global batch, size, seed, world)
    'ddp_unet_resnet18_w2': ('unet', 'resnet18', 2, 4, 64, 130, 2),
    'ddp_linknet_resnet50_w4': ('linknet', 'resnet50', 2, 8, 64, 131, 4),
}
# cases larger than this many logits store a centre crop + checksums instead of the full tensor
FULL_LOGITS_MAX = 64 * 1024


def case_batch(B, classes, S, seed, arch=None):
    """Synthetic batch of a case; a (H, W) size is cut out of the square frame of side max(H, W).  DeepLabV3+ frames get a brightness of
    their own each: its pooled ASPP branch normalises B values per channel, and BatchNorm of B nearly equal numbers amplifies rounding
    without bound (tests/test_gpu_deeplab.py)."""
    from synth import make_batch
    if isinstance(S, tuple):
        H, W = S
        img, mask = make_batch(B, classes, max(H, W), seed=seed, empty_last=(classes > 1))
        img, mask = img[:, :, :H, :W].contiguous(), mask[:, :, :H, :W].contiguous()
    else:
        img, mask = make_batch(B, classes, S, seed=seed, empty_last=(classes > 1))
    if arch in ('deeplabv3plus', 'pspnet', 'deeplabv3'):
        img = (img * (0.35 + 0.65 * torch.arange(B).view(B, 1, 1, 1) / max(1, B - 1))).round().contiguous()
    return img, mask


def case_keep(arch, B, S, seed):
    """Dropout keep pattern of a case in torch's layout: FPN Dropout2d [B, 128] (p = 0.2), DeepLabV3+ element-wise [B, 256, H/16, W/16]
    (p = 0.5), PSPNet Dropout2d [B, 512] (p = 0.2); None for the architectures without dropout."""
    H, W = S if isinstance(S, tuple) else (S, S)
    g = torch.Generator().manual_seed(seed + 2)
    if arch == 'fpn':
        return (torch.rand(B, 128, generator=g) < 0.8).float()
    if arch == 'pspnet':
        return (torch.rand(B, 512, generator=g) < 0.8).float()
    if arch in ('deeplabv3plus', 'deeplabv3'):
        s_ = 16 if arch == 'deeplabv3plus' else 8
        return (torch.rand(B, 256, H // s_, W // s_, generator=g) < 0.5).float()
    return None


def summarize_logits(z):
    """What is stored of a logits array [B, C, H, W]: all of it, or (large cases) a 32 x 32 centre crop of every
    channel plus float64 sum, sum of squares and sum of |z|."""
    z = np.asarray(z, dtype=np.float32)
    if z.size <= FULL_LOGITS_MAX:
        return {'logits': z}
    H, W = z.shape[2], z.shape[3]
    crop = z[:, :, H // 2 - 16:H // 2 + 16, W // 2 - 16:W // 2 + 16].copy()
    z64 = z.astype(np.float64)
    return {'logits_crop': crop, 'logits_sums': np.array([z64.sum(), (z64 * z64).sum(), np.abs(z64).sum()]),
            'logits_absmax': np.float64(np.abs(z64).max())}


def build(arch, enc, classes, seed, kinkfree=True):
    from oracle import create_model
    from oracle.nets import randomize_bn
    torch.manual_seed(seed)
    m = create_model(arch, enc, classes=classes)
    randomize_bn(m, seed)
    if kinkfree:  # BN biases at +-8: gradients are smooth in the fp32 rounding (see tests/test_gpu_net.py)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.bias.copy_(8.0 * ((torch.rand(mod.bias.shape, generator=g) < 0.7).float() * 2 - 1))
            if arch in ('fpn', 'deeplabv3plus', 'pspnet', 'deeplabv3'):
                # their heads sit straight behind normalisation layers with +-8 biases (summed four times in FPN): keep |logits| of order 1,
                # a saturated sigmoid has no gradient to compare
                for mod in m.modules():
                    if isinstance(mod, torch.nn.GroupNorm):
                        mod.bias.copy_(8.0 * ((torch.rand(mod.bias.shape, generator=g) < 0.7).float() * 2 - 1))
                m.segmentation_head[0].weight.mul_(0.03)
    return m.train()


def run_case(arch, enc, classes, B, S, seed):
    from oracle import DiceLoss, get_stats
    m = build(arch, enc, classes, seed)
    img, mask = case_batch(B, classes, S, seed, arch)
    keep = case_keep(arch, B, S, seed)
    if keep is not None:
        m.decoder.dropout.mask = keep
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    logits = m((img - mean) / std)
    loss = DiceLoss()(logits, mask)
    loss.backward()
    tp, fp, fn, tn = get_stats((logits.detach().sigmoid() > 0.5).long(), mask.long())
    return {'logits': logits.detach().numpy(), 'loss': float(loss.item()),
            'grad_abs_sums': np.array([0.0 if p.grad is None else p.grad.abs().sum().item() for _, p in m.named_parameters()], dtype=np.float64),
            'stats': torch.stack([tp, fp, fn, tn], dim=-1).numpy()}


def shard_range(n_items, rank, world):
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_ddp_case(arch, enc, classes, B, S, seed, world):
    """Sequential emulation of `world` DDP ranks on the oracle."""
    from oracle import DiceLoss, get_stats
    img, mask = case_batch(B, classes, S, seed)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    losses, stats, acc, rank0_buffers = [], [], None, None
    for r in range(world):
        m = build(arch, enc, classes, seed)          # every rank starts from rank 0's parameters and buffers
        lo, hi = shard_range(B, r, world)
        logits = m((img[lo:hi] - mean) / std)
        loss = DiceLoss()(logits, mask[lo:hi])
        loss.backward()
        g = [p.grad / world for _, p in m.named_parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        losses.append(float(loss.item()))
        tp, fp, fn, tn = get_stats((logits.detach().sigmoid() > 0.5).long(), mask[lo:hi].long())
        stats.append(torch.stack([tp, fp, fn, tn], dim=-1).numpy())
        if r == 0:
            rank0_buffers = np.array([[b.double().sum().item(), b.double().abs().sum().item()] for n, b in m.named_buffers()
                                      if n.endswith('running_mean') or n.endswith('running_var')])
    return {'losses': np.array(losses), 'stats': np.concatenate(stats, axis=0),
            'grad_abs_sums': np.array([a.abs().sum().item() for a in acc], dtype=np.float64),
            'grad_sums': np.array([a.double().sum().item() for a in acc], dtype=np.float64),
            'rank0_buffer_sums': rank0_buffers}


if __name__ == '__main__':
    for name, case in DDP_CASES.items():
        if os.path.exists(os.path.join(HERE, f'{name}.npz')) and '--all' not in sys.argv:
            continue
        out = run_ddp_case(*case)
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
        print(name, 'losses', out['losses'])
    for name, case in CASES.items():
        if os.path.exists(os.path.join(HERE, f'{name}.npz')) and '--all' not in sys.argv:
            continue   # committed vectors are only regenerated on request
        out = run_case(*case)
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), loss=np.float64(out['loss']), grad_abs_sums=out['grad_abs_sums'],
                            stats=out['stats'], **summarize_logits(out['logits']))
        print(name, 'loss', out['loss'], 'logits', out['logits'].shape)
