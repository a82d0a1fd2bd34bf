"""GPU box: what HBM bandwidth plain streaming kernels reach on this card (the yardstick for the BatchNorm sweeps).
copy = torch device-to-device copy (read + write), read = torch.sum over bf16 (read only), at three sizes."""
import torch
for mb in (64, 256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device='cuda', dtype=torch.bfloat16)
    b = torch.empty_like(a)
    for name, fn, nbytes in (('copy', lambda: b.copy_(a), 2 * a.numel() * 2), ('read', lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32), a.numel() * 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'{name} {mb} MiB tensor: {ms * 1e3:.1f} us  {nbytes / ms / 1e9:.2f} TB/s')
