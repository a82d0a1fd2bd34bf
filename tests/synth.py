"""Seeded synthetic "OCT-shaped" batches: BGR float32 images in 0..255 with
a dark background outside the inscribed circle, and {0,1} float32 masks with an optionally empty
class (exercises the Dice empty-class branch)."""
import torch


def make_batch(B, C, S, seed=1234, empty_last=False):
    g = torch.Generator().manual_seed(seed)
    r = (56.0 + 52.0 * torch.randn(B, 1, S, S, generator=g)).clamp(0, 255)
    r = torch.nn.functional.avg_pool2d(r, 5, 1, 2)
    img = torch.cat([r * 0.05, r * 0.37, r], dim=1).round()
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing='ij')
    rad2 = ((yy - S / 2 + 0.5) ** 2 + (xx - S / 2 + 0.5) ** 2).float()
    img = img * (rad2 <= (S / 2) ** 2).float()
    masks = []
    for c in range(C):
        cy = S / 2 + (torch.rand(B, generator=g) - 0.5) * S / 8
        cx = S / 2 + (torch.rand(B, generator=g) - 0.5) * S / 8
        area = 0.10 + 0.20 * torch.rand(B, generator=g)
        rr = (area * S * S / 3.14159).sqrt() / (1 + c)
        m = (((yy[None] - cy[:, None, None]) ** 2 + (xx[None] - cx[:, None, None]) ** 2) <= rr[:, None, None] ** 2).float()
        masks.append(m)
    mask = torch.stack(masks, dim=1)
    if empty_last and C > 1:
        mask[:, -1] = 0
    return img.contiguous(), mask.contiguous()
