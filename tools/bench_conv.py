"""Micro-benchmark of one conv shape through the single-op C ABI (GPU box).
usage: bench_conv.py N H W Cin Cout R [stride] [mode=fwd|dgrad|wgrad] [iters]        (TR=1: ConvTranspose2d k4 s2 p1, H x W = its input)"""
import sys
sys.path.insert(0, '.')
import torch
import os
if os.environ.get('OCTSEG_LIB'):   # an experimental build of the library (timing experiments)
    from oct_segmentation_amd import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ['OCTSEG_LIB'])
from oct_segmentation_amd import ops

N, H, W, Cin, Cout, R = map(int, sys.argv[1:7])
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
mode = sys.argv[8] if len(sys.argv) > 8 else 'fwd'
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 10
pad = R // 2
import os as _os
tr = _os.environ.get('TR') == '1'
if tr:
    pad = 1
dev = 'cuda'
x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
w = torch.randn(R, R, Cout, Cin, device=dev) * 0.05
OH = H * 2 if tr else (H + 2 * pad - R) // stride + 1
dy = torch.randn(N, OH, OH, Cout, device=dev).bfloat16()
import os
if os.environ.get('DATA') == 'zeros':      # DVFS probe: no operand toggling in the MFMA pipe
    x.zero_(); w.zero_(); dy.zero_()
elif os.environ.get('DATA') == 'ones':
    x.fill_(1); w.fill_(1); dy.fill_(1)
def run():
    if mode == 'fwd':
        return ops.conv2d_forward(x, w, None, stride, pad, tr)
    if mode == 'dgrad':
        return ops.conv2d_backward_data(dy, w, (H, W), stride, pad, tr)
    return ops.conv2d_backward_weight(x, dy, R, stride, pad, tr)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * N * OH * OH * Cout * Cin * (4 if tr else R * R)
print(f'{mode} N{N} {H}x{W} {Cin}->{Cout} k{R} s{stride}: {ms:.3f} ms/iter (incl. weight pack) {fl / ms / 1e9:.1f} TF/s')
