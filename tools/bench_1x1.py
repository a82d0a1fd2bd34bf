"""GPU box: stride-1 1x1 conv shapes through the single-op C ABI, HIP-event timing of the conv launch alone
(the weight pack runs once, outside the timed region, by timing forward twice with and without... no: see below).
usage: bench_1x1.py"""
import sys
sys.path.insert(0, '.')
import ctypes as C
import torch
from oct_segmentation_amd import _lib as L

def time_fwd(N, H, W, Cin, Cout, iters=20, aff=False):
    dev = 'cuda'
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    w = (torch.randn(1, 1, Cout, Cin, device=dev) * 0.05).contiguous()
    y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    lib = L.lib()
    sc = torch.empty(lib.octseg_conv2d_scratch_bytes(L.BF16, N, H, W, Cin, Cout, 1, 1), dtype=torch.uint8, device=dev)
    def run():
        L.check(lib.octseg_conv2d_forward(L.BF16, L.ptr(x), L.ptr(w), None, L.ptr(y), N, H, W, Cin, Cout, 1, 1, 1, 0, 0, L.ptr(sc), L.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    # pack-only time: a tiny problem with the same weights
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e3
    x1 = x[:1, :2, :8].contiguous(); y1 = torch.empty(1, 2, 8, Cout, device=dev, dtype=torch.bfloat16)
    def run1():
        L.check(lib.octseg_conv2d_forward(L.BF16, L.ptr(x1), L.ptr(w), None, L.ptr(y1), 1, 2, 8, Cin, Cout, 1, 1, 1, 0, 0, L.ptr(sc), L.stream_ptr()))
    for _ in range(3): run1()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): run1()
    e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1) / iters * 1e3
    M = N * H * W
    byts = M * Cin * 2 + M * Cout * 2
    print(f'{N}x{H}x{W} {Cin:5d}->{Cout:5d}: {t:7.1f} us (tiny problem with the same pack: {t1:5.1f} us) -> ~{t - t1 + 8:6.1f} us;  {byts / 1e6:6.1f} MB algorithmic = {byts / (t - t1 + 8) / 1e6:5.2f} TB/s, {2.0 * M * Cin * Cout / (t - t1 + 8) / 1e6:6.1f} TF/s')

for shape in [(16, 44, 44, 1024, 32), (16, 44, 44, 1024, 64), (16, 44, 44, 1024, 128), (16, 44, 44, 1024, 256), (16, 44, 44, 256, 1024),
              (16, 176, 176, 256, 64), (16, 176, 176, 64, 256), (16, 88, 88, 512, 128), (16, 88, 88, 128, 512), (16, 22, 22, 2048, 512), (16, 22, 22, 512, 2048)]:
    time_fwd(*shape)
