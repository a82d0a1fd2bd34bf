"""Debug helper (GPU box): per-parameter gradient error of the engine vs the oracle."""
import sys
import torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from synth import make_batch
from test_gpu_net import _oracle, _relmax
from oct_segmentation_amd.engine import SegNet
from oracle import DiceLoss

arch, enc, classes, B, S = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dt = torch.float32 if len(sys.argv) < 7 else torch.bfloat16
ref = _oracle(arch, enc, classes)
net = SegNet(arch, enc, classes=classes, device='cuda', compute_dtype=dt)
net.load_state_dict(ref.state_dict())
img, mask = make_batch(B, classes, S, seed=11, empty_last=(classes > 1))
ref.train()
lr = ref(img)
loss_ref = DiceLoss()(lr, mask)
loss_ref.backward()
net.train()
loss, logits, stats = net.train_step_raw(img.cuda(), mask.cuda())
torch.cuda.synchronize()
g = net.named_grads()
for n, p in ref.named_parameters():
    a, b = g[n].cpu(), p.grad
    cos = float((a.flatten() @ b.flatten()) / (a.norm() * b.norm() + 1e-30))
    print(f'{n:55s} rel={_relmax(a, b):.3e} cos={cos:.5f} |ref|={b.abs().max().item():.3e} |got|={a.abs().max().item():.3e}')
