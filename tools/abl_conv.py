"""Diagnostic (GPU box): time one conv shape with a library built from a given conv source + extra -D flags.
usage: abl_conv.py <conv_source.hip> "<flags>" N H W Cin Cout R"""
import os, subprocess, sys, shutil, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'oct_segmentation_amd', 'csrc')
src, flags = sys.argv[1], sys.argv[2].split()
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, 'lib.so')
srcs = [src] + [os.path.join(csrc, f) for f in ('wgrad_mfma.hip', 'elementwise.hip', 'augment.hip', 'plan.cpp')]
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-w', '-I', csrc, '-x', 'hip'] + flags + ['-o', so] + srcs, check=True)
sys.path.insert(0, root)
import torch
from oct_segmentation_amd import _lib as L
L.LIB_PATH = so
from oct_segmentation_amd import ops
N, H, W, Cin, Cout, R = map(int, sys.argv[3:9])
x = torch.randn(N, H, W, Cin, device='cuda').bfloat16()
w = torch.randn(R, R, Cout, Cin, device='cuda') * 0.05
for _ in range(3): ops.conv2d_forward(x, w, None, 1, R // 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.conv2d_forward(x, w, None, 1, R // 2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f'{os.path.basename(src)} {flags}: {ms:.3f} ms  {2.0*N*H*W*Cout*Cin*R*R/ms/1e9:.1f} TF/s')
shutil.rmtree(tmp, ignore_errors=True)
