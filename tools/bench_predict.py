"""GPU box: latency of the serving path (eval forward, reference model.py:183-200 / predict.py segment()) with and
without the hipGraph replay.  usage: bench_predict.py [arch] [encoder] [B] [H]"""
import sys, time
sys.path.insert(0, '.')
import torch
from oct_segmentation_amd.engine import SegNet
arch = sys.argv[1] if len(sys.argv) > 1 else 'unetplusplus'
enc = sys.argv[2] if len(sys.argv) > 2 else 'resnet101'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H = int(sys.argv[4]) if len(sys.argv) > 4 else 704
net = SegNet(arch, enc, classes=1, device='cuda', compute_dtype=torch.bfloat16, seed=0).eval()
x = torch.rand(B, 3, H, H, device='cuda') * 255
for mode in (False, True):
    net.use_graph = mode
    for _ in range(4):
        y = net(x)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        y = net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{arch}/{enc} B={B} {H}x{H} bf16 eval forward, {"hipGraph replay" if mode else "eager launches "}: {dt * 1e3:.3f} ms/call, {B / dt:.1f} frames/s')
