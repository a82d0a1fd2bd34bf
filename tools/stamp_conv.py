"""Diagnostic (GPU box): build the library with -DOCTSEG_STAMP into a temp dir and print where the conv tap
loop spends its cycles for one shape.  usage: stamp_conv.py N H W Cin Cout R"""
import os, subprocess, sys, shutil, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'oct_segmentation_amd', 'csrc')
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, 'liboctseg_stamp.so')
srcs = [os.path.join(csrc, f) for f in ('conv_mfma.hip', 'conv3x3p.hip', 'gemm1x1.hip', 'wgrad_mfma.hip', 'elementwise.hip', 'augment.hip', 'plan.cpp')]
prebuilt = os.path.join(root, 'oct_segmentation_amd', 'liboctseg_stamp.so')   # built in the build container, travels with the snapshot
if os.path.exists(prebuilt):
    so = prebuilt
else:
  subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-w', '-DOCTSEG_STAMP'] + os.environ.get('OCTSEG_EXTRA_DEFS', '').split() + ['-o', so] + srcs, check=True)
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
L.check(L.lib().octseg_debug_set_stamp(None))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv2d_forward(x, w, None, 1, R // 2)
e1.record(); torch.cuda.synchronize()
print(f'stamped build: {e0.elapsed_time(e1) / 5:.3f} ms/iter')
print(f'{N}x{H}x{W} {Cin}->{Cout} k{R}')
n = max(1, b[5])
names = ['issue (slab DMA, slice load)', 'MFMA block', 'vmcnt wait', 'affine + LDS store + lgkm', 'barrier']
tot = max(1, sum(b[:5]))
for i in range(5):
    print(f'  {names[i]:42s} {b[i] / n:8.1f} ticks/tap  {100.0 * b[i] / tot:5.1f} %')
print(f'  total {tot / n:.1f} ticks per wave-tap ({n} wave-taps)')
if b[11]:
    print(f'  per wave: prologue {b[8] / b[11]:.0f}  main loop {b[9] / b[11]:.0f}  epilogue {b[10] / b[11]:.0f} ticks ({b[11]} waves)')
shutil.rmtree(tmp, ignore_errors=True)
