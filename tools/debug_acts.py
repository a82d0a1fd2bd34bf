"""Debug helper (GPU box): per-conv raw output and dy error of the engine vs the oracle."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from synth import make_batch
from test_gpu_net import _oracle, _relmax
from oct_segmentation_amd.engine import SegNet, debug_tensor
from oracle import DiceLoss

arch, enc, classes, B, S = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ref = _oracle(arch, enc, classes)
net = SegNet(arch, enc, classes=classes, device='cuda', compute_dtype=torch.float32)
net.load_state_dict(ref.state_dict())
img, mask = make_batch(B, classes, S, seed=11, empty_last=(classes > 1))
acts, grads = {}, {}
def mk(name):
    def hook(m, i, o):
        acts[name] = o.detach()
        o.register_hook(lambda g: grads.__setitem__(name, g.detach()))
    return hook
for n, m in ref.named_modules():
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and 'segmentation_head' not in n:
        m.register_forward_hook(mk(n))
ref.train()
lr = ref(img)
DiceLoss()(lr, mask).backward()
net.train()
loss, logits, stats = net.train_step_raw(img.cuda(), mask.cuda())
torch.cuda.synchronize()
plan = net._plan(B, S, S)
for n in acts:
    y = debug_tensor(net, plan, n).cpu()
    dy = debug_tensor(net, plan, n, grad=True).cpu()
    print(f'{n:45s} y rel={_relmax(y, acts[n]):.2e}  dy rel={_relmax(dy, grads[n]):.2e} |dy|={grads[n].abs().max().item():.2e} shape={tuple(y.shape)}')
    e = (dy - grads[n]).abs()
    thr = 1e-3 * grads[n].abs().max()
    bad = (e > thr).nonzero()
    if len(bad):
        print('   bad elements:', len(bad), 'of', e.numel(), 'first:', bad[:6].tolist(), 'last:', bad[-3:].tolist())
        ys = sorted(set(bad[:, 2].tolist())); xs = sorted(set(bad[:, 3].tolist())); cs = sorted(set(bad[:, 1].tolist()))
        print('   rows', ys[:20], 'cols', xs[:20], 'chans', cs[:20], len(cs))
