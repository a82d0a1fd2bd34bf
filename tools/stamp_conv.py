"""Diagnostic (GPU box): build the library with -DOCTSEG_STAMP into a temp dir and print where the conv tap
loop spends its cycles for one shape.  usage: stamp_conv.py N H W Cin Cout R"""
import os, subprocess, sys, shutil, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'oct_segmentation_amd', 'csrc')
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, 'liboctseg_stamp.so')
srcs = [os.path.join(csrc, f) for f in ('conv_mfma.hip', 'wgrad_mfma.hip', 'elementwise.hip', 'plan.cpp')]
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-w', '-DOCTSEG_STAMP', '-o', so] + srcs, check=True)
sys.path.insert(0, root)
import torch
from oct_segmentation_amd import _lib as L
L.LIB_PATH = so
from oct_segmentation_amd import ops
N, H, W, Cin, Cout, R = map(int, sys.argv[1:7])
x = torch.randn(N, H, W, Cin, device='cuda').bfloat16()
w = torch.randn(R, R, Cout, Cin, device='cuda') * 0.05
ops.conv2d_forward(x, w, None, 1, R // 2)
buf = torch.zeros(16, dtype=torch.int64, device='cuda')
L.check(L.lib().octseg_debug_set_stamp(L.ptr(buf)))
ops.conv2d_forward(x, w, None, 1, R // 2)
torch.cuda.synchronize()
b = buf.cpu().tolist()
print(f'{N}x{H}x{W} {Cin}->{Cout} k{R}')
nc = max(1, b[2]); print(f'consumer wave-iterations {nc}: MFMA block {b[0]/nc:.0f} cyc, barrier wait {b[1]/nc:.0f} cyc')
na = max(1, b[6]); print(f'producer acting turns {na}: finish(wait+affine+store) {b[3]/na:.0f} cyc, issue {b[4]/na:.0f} cyc, barrier wait {b[5]/na:.0f} cyc')
print(f'  of which setup() per acting turn: {b[7]/na:.0f} cyc')
shutil.rmtree(tmp, ignore_errors=True)
