import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from oct_segmentation_amd.engine import SegNet
dev = torch.device('cuda:0')
enc = sys.argv[2] if len(sys.argv) > 2 else 'resnet18'
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfgs = [('unetplusplus', enc, 1), ('linknet', enc, 2), ('unet', enc, 1)]
order = sys.argv[1] if len(sys.argv) > 1 else 'interleaved'
nets = [SegNet(a, e, classes=c, device=dev, compute_dtype=torch.float16, seed=i).eval() for i, (a, e, c) in enumerate(cfgs)]
for n in nets: n.use_graph = True
x = torch.rand(1, 3, S, S, device=dev) * 255
try:
    if order == 'interleaved':
        for step in range(4):
            for i, n in enumerate(nets):
                print('step', step, 'net', i, flush=True)
                z = n(x)
                if len(sys.argv) > 4:
                    from oct_segmentation_amd import _lib as L
                    out = torch.zeros(1, S, S, 4, device=dev)
                    L.check(L.lib().octseg_mask_assemble(L.ptr(z), 1, z.shape[1], S, S, 0, L.ptr(out), S, S, 4, 0, None, None, L.stream_ptr()))
    else:
        for i, n in enumerate(nets):
            for step in range(4):
                print('net', i, 'step', step, flush=True)
                n(x)
    torch.cuda.synchronize()
    print('ok')
except Exception as e:
    print('FAILED:', e)
