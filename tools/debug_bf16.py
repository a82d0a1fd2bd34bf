"""Debug helper (GPU box): global gradient cosine of the bf16 / fp32 engine vs the fp32 oracle at several sizes."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from synth import make_batch
from test_gpu_net import _oracle
from oct_segmentation_amd.engine import SegNet
from oracle import DiceLoss

for arch, enc, classes, B, S in [('unet', 'resnet18', 1, 2, 64), ('unet', 'resnet18', 1, 4, 128), ('unet', 'resnet18', 1, 4, 256),
                                 ('linknet', 'resnet18', 2, 4, 128), ('linknet', 'resnet18', 2, 4, 256), ('unetplusplus', 'resnet18', 1, 4, 128)]:
    ref = _oracle(arch, enc, classes)
    img, mask = make_batch(B, classes, S, seed=11)
    ref.train()
    loss_ref = DiceLoss()(ref(img), mask)
    loss_ref.backward()
    for dt in (torch.float32, torch.bfloat16):
        net = SegNet(arch, enc, classes=classes, device='cuda', compute_dtype=dt)
        net.load_state_dict({k: v for k, v in _oracle(arch, enc, classes).state_dict().items()})
        net.train()
        loss, logits, stats = net.train_step_raw(img.cuda(), mask.cuda())
        torch.cuda.synchronize()
        g = net.named_grads()
        num = da = db = 0.0
        worst = (1.0, '')
        for n, p in ref.named_parameters():
            a, b = g[n].cpu().flatten().double(), p.grad.flatten().double()
            num += float(a @ b); da += float(a @ a); db += float(b @ b)
            c = float(a @ b) / (float(a.norm() * b.norm()) + 1e-30)
            if b.norm() > 1e-8 and c < worst[0]:
                worst = (c, n)
        print(f'{arch} {enc} B{B} S{S} {dt}: loss {loss.item():.6f} ref {loss_ref.item():.6f} cos {num / (da ** .5 * db ** .5):.6f} worst-param cos {worst[0]:.4f} {worst[1]}', flush=True)
