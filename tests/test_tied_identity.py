"""The identity behind the tied decomposition (plan.h ConvLayer::tie, kernels.h PackJob::tied), pinned in float64 on the CPU: a 3x3 conv over a
nearest-x2 upsampled map (smp DecoderBlock: F.interpolate(scale_factor=2, mode='nearest') + Conv2dReLU, the graph src/models/smp/model.py:65-71
builds) equals ConvTranspose2d(k4, s2, p1) of the low-resolution map with K4[u][v] = sum of W3[r][s] over r in A(u), s in A(v),
A = {0: [2], 1: [1, 2], 2: [0, 1], 3: [0]} -- border rows included -- and the gradient fold is the transpose of that sum."""
import torch


def test_tied_kernel_and_fold_against_float64():
    """ConvTranspose2d(k4, s2, p1) with K4[u][v] = sum of W3 over A(u) x A(v) == conv3x3(nearest x2 (a)) in float64 on the CPU (the identity the
    engine relies on, border rows included), and the fold is its transpose."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w3 = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)          # torch layout [O][I][3][3]
    A = {0: [2], 1: [1, 2], 2: [0, 1], 3: [0]}
    k4 = torch.zeros(4, 5, 4, 4, dtype=torch.float64)
    for u in range(4):
        for v in range(4):
            for r in A[u]:
                for s in A[v]:
                    k4[:, :, u, v] += w3[:, :, r, s]
    up = torch.nn.functional.interpolate(a, scale_factor=2, mode='nearest')
    want = torch.nn.functional.conv2d(up, w3, padding=1)
    # conv_transpose2d correlates with the flipped kernel: out[2y + py] = sum_r K[r] a[y + (py + 1 - r) / 2] is torch's rule with weight [I][O][u][v]
    got = torch.nn.functional.conv_transpose2d(a, k4.permute(1, 0, 2, 3), stride=2, padding=1)
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
    # fold: d(loss)/dW3[r][s] = sum of dK4[u][v] over the (u, v) with r in A(u), s in A(v)
    dk4 = torch.randn(4, 5, 4, 4, generator=g, dtype=torch.float64)
    dw3 = torch.zeros_like(w3)
    for u in range(4):
        for v in range(4):
            for r in A[u]:
                for s in A[v]:
                    dw3[:, :, r, s] += dk4[:, :, u, v]
    for r in range(3):
        for s in range(3):
            us, vs = [2 - r, 3 - r], [2 - s, 3 - s]          # what tied_fold_kernel sums
            assert torch.equal(dw3[:, :, r, s], sum(dk4[:, :, u, v] for u in us for v in vs))
