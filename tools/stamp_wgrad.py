"""Diagnostic (GPU box): per-phase cycles of the wgrad tile loop (library built with -DOCTSEG_STAMP)."""
import os, subprocess, sys, shutil, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'oct_segmentation_amd', 'csrc')
tmp = tempfile.mkdtemp(); so = os.path.join(tmp, 'lib.so')
srcs = [os.path.join(csrc, f) for f in ('conv_mfma.hip', 'conv3x3p.hip', 'gemm1x1.hip', 'wgrad_mfma.hip', 'elementwise.hip', 'augment.hip', 'plan.cpp')]
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-w', '-DOCTSEG_STAMP'] + os.environ.get('OCTSEG_EXTRA_DEFS', '').split() + ['-o', so] + srcs, check=True)
sys.path.insert(0, root)
import torch
from oct_segmentation_amd import _lib as L
L.LIB_PATH = so
from oct_segmentation_amd import ops
N, H, W, Cin, Cout, R = map(int, sys.argv[1:7])
x = torch.randn(N, H, W, Cin, device='cuda').bfloat16()
dy = torch.randn(N, H, W, Cout, device='cuda').bfloat16()
ops.conv2d_backward_weight(x, dy, R, 1, R // 2)
buf = torch.zeros(16, dtype=torch.int64, device='cuda')
L.check(L.lib().octseg_debug_set_stamp(L.ptr(buf)))
ops.conv2d_backward_weight(x, dy, R, 1, R // 2)
torch.cuda.synchronize()
b = buf.cpu().tolist(); n = max(1, b[4]); tot = sum(b[:4])
print(f'wgrad {N}x{H}x{W} {Cin}->{Cout} k{R}: {n} wave-tiles, {tot/n:.0f} cycles per tile per wave')
for nm, v in zip(['issue next-tile loads', 'MFMA (tr reads + mfma)', 'barrier 1', 'wait+affine+LDS stores+barrier 2'], b[:4]):
    print(f'  {nm:34s} {v/n:8.1f} cycles {100.0*v/tot:5.1f} %')
shutil.rmtree(tmp, ignore_errors=True)
