"""GPU parity of the conv kernel families against torch CPU (fp32 reference).

Each case runs forward, data gradient and weight gradient through the C ABI
single-op entry points and compares with torch.nn.functional on the CPU.
Tolerances: fp32 path 1e-4 relative to the output scale (BASELINE north_star);
bf16 path 2e-2 (bf16 inputs, f32 accumulate).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, R, stride, pad, transposed)
CASES = [
    (2, 16, 16, 64, 64, 3, 1, 1, False),
    (1, 16, 32, 64, 128, 3, 1, 1, False),
    (2, 24, 40, 32, 16, 3, 1, 1, False),     # ragged tiles, Cin < chunk, small Cout
    (1, 8, 8, 192, 256, 3, 1, 1, False),     # multi-chunk K, wide N
    (2, 16, 16, 64, 128, 1, 1, 0, False),    # 1x1
    (2, 16, 16, 128, 64, 1, 2, 0, False),    # 1x1 stride 2 (downsample)
    (2, 32, 32, 64, 64, 3, 2, 1, False),     # 3x3 stride 2
    (2, 8, 16, 64, 64, 4, 2, 1, True),       # ConvTranspose2d k4 s2 p1
    (1, 12, 20, 16, 32, 4, 2, 1, True),
    (1, 16, 16, 160, 64, 1, 1, 0, False),    # stem GEMM shape (K = 160)
    # stride-1 1x1 layers: the persistent GEMM kernel (gemm1x1.hip) in bf16 where K % 64 == 0, conv_mfma_kernel otherwise / in f32
    (3, 7, 5, 128, 40, 1, 1, 0, False),      # ragged M tile (105 pixels), N tile of 64 with 40 real channels
    (2, 12, 12, 256, 320, 1, 1, 0, False),   # three N tiles, the last one half full; dgrad contracts over 320 channels (5 K steps)
    (1, 44, 44, 1024, 256, 1, 1, 0, False),  # ResNet-101 layer3 conv1: 16 M tiles, 16 K steps
    (4, 22, 22, 256, 1024, 1, 1, 0, False),  # layer3 conv3: 8 N tiles
    (2, 24, 24, 64, 16, 1, 1, 0, False),     # LinkNet decoder tail: one 32-wide N tile, one K step
    (8, 96, 96, 64, 64, 1, 1, 0, False),     # 576 M tiles > 512 workgroups: some walk two tiles (pipeline across the tile boundary)
    (5, 96, 96, 128, 256, 1, 1, 0, False),   # 360 M tiles x 2 N tiles on 256 workgroup rows, two K steps, ragged walk (104 rows do two tiles)
    # persistent 3x3 kernel (conv3x3p.hip, bf16): 512 tiles x 2 N tiles over 256 workgroups = four tiles each, three K chunks with a
    # partial last one (136 = 64 + 64 + 8), a half-empty second N tile; forward and data gradient both walk tiles
    (8, 128, 128, 160, 136, 3, 1, 1, False),
    # 256 -> 256 on a 16-divisible map: routed to the persistent kernel instead of the wide-N tile (512 items = two per workgroup)
    (4, 128, 128, 256, 256, 3, 1, 1, False),
    # the wide-N tile (256 output channels, K >= 256, >= 1024 workgroups) where the persistent kernel is not eligible: a 120 x 120 map
    # (7.5 tiles per side: ragged tiles in both directions)
    (16, 120, 120, 256, 256, 3, 1, 1, False),
    # thin full-resolution layers (thin.hip, bf16; f32 stays on conv_mfma_kernel): 16 / 32 channels in, 16 / 32 out on >= 4096 pixels --
    # weights as the MFMA A operand, persistent 32-pixel-wide tiles; ragged widths (72 = 2 x 32 + 8, 104 = 3 x 32 + 8) and heights
    (2, 64, 96, 16, 16, 3, 1, 1, False),     # x_0_4.conv2 shape
    (2, 64, 96, 32, 16, 3, 1, 1, False),     # x_0_4.conv1 shape (8-row tiles, 96-byte LDS pixels); its dgrad writes 32 channels
    (1, 80, 72, 16, 32, 3, 1, 1, False),
    (3, 40, 104, 32, 32, 3, 1, 1, False),    # x_0_3.conv2 shape: two output blocks both ways
    (5, 112, 96, 16, 16, 3, 1, 1, False),    # 1050 tiles > 768 workgroups: some walk two tiles
    # thin 1x1 layers (LinkNet's last decoder block / head shapes) and the parity launches of a thin ConvTranspose2d forward
    (2, 64, 96, 16, 32, 1, 1, 0, False),     # forward 16 -> 32 (two output blocks), dgrad 32 -> 16, wgrad with a 32-channel dy
    (2, 64, 104, 32, 16, 1, 1, 0, False),
    (3, 40, 72, 32, 32, 1, 1, 0, False),
    (2, 48, 64, 16, 16, 4, 2, 1, True),      # ConvTranspose2d 16 -> 16: four 2x2-tap launches writing at stride 2
    (1, 64, 80, 32, 32, 4, 2, 1, True),
    # DeepLabV3+ shapes: the pooled ASPP branch is a 1x1 conv on a 1x1 map (M = batch pixels), the separable convs' pointwise halves
    # contract over 304 channels (4.75 K steps) and the high-resolution branch has 48 output channels
    (2, 1, 1, 512, 256, 1, 1, 0, False),
    (4, 1, 1, 2048, 256, 1, 1, 0, False),
    (2, 16, 24, 304, 256, 1, 1, 0, False),
    (2, 16, 24, 256, 48, 1, 1, 0, False),
    # the LDS-DMA ring weight gradient (wgrad1x1.hip, bf16; >= 1024 pixels, 64-divisible channels): 1200 pixels = 18.75 stages of 64 (a
    # three-k-step tail), three ci tiles, the 64-channel wave tile; and the 128-channel tile on two dy planes with split-K ranges of 4 stages
    (3, 20, 20, 192, 64, 1, 1, 0, False),
    (2, 32, 32, 64, 128, 1, 1, 0, False),
    (1, 64, 80, 128, 384, 1, 1, 0, False),
    # ConvTranspose2d weight gradient with all four output parities in one launch (wgrad_convt.hip, bf16): ragged 8 x 16 tiles of the
    # low-resolution map (12 x 20), 1.5 ci tiles x 2.5 co tiles; and 60 tiles walked by split-K workgroups (window / plane double buffers)
    (2, 12, 20, 96, 160, 4, 2, 1, True),
    (4, 40, 48, 128, 64, 4, 2, 1, True),
    # 11 x 11 pixel tiles (conv_mfma.hip LOOP_T11, bf16): maps that are multiples of 11 and not of 16 -- the 88^2 / 44^2 / 22^2 maps of a 704^2 frame;
    # forward and data gradient (> 64 channels on the output side of either), a half-filled second N tile, a non-square map, one tile per image
    (2, 44, 44, 128, 256, 3, 1, 1, False),
    (3, 22, 22, 192, 128, 3, 1, 1, False),
    (1, 88, 88, 128, 128, 3, 1, 1, False),
    (2, 33, 55, 128, 192, 3, 1, 1, False),
    (2, 11, 11, 256, 128, 3, 1, 1, False),
]


def _ref(x, w, stride, pad, transposed):
    if transposed:
        return F.conv_transpose2d(x, w, stride=stride, padding=pad)
    return F.conv2d(x, w, stride=stride, padding=pad)


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_family(cuda, case, dtype):
    from oct_segmentation_amd import ops
    N, H, W, Cin, Cout, R, stride, pad, tr = case
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(N, Cin, H, W, generator=g)
    wshape = (Cin, Cout, R, R) if tr else (Cout, Cin, R, R)
    w = torch.randn(wshape, generator=g) / (Cin * R * R) ** 0.5
    if dtype == torch.bfloat16:  # quantise the inputs so the reference sees the same operands
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = _ref(x, w, stride, pad, tr)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)
    tol = 1e-4 if dtype == torch.float32 else 2e-2

    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(cuda, dtype)
    wa = ops.weight_to_arena(w, tr).to(cuda)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(cuda, dtype)

    yd = ops.conv2d_forward(xd, wa, None, stride, pad, tr)
    torch.cuda.synchronize()
    e_fwd = _rel(yd.float().cpu().permute(0, 3, 1, 2), y.detach())

    dxd = ops.conv2d_backward_data(dyd, wa, (H, W), stride, pad, tr)
    torch.cuda.synchronize()
    e_dx = _rel(dxd.float().cpu().permute(0, 3, 1, 2), x.grad)

    dwd = ops.conv2d_backward_weight(xd, dyd, R, stride, pad, tr)
    torch.cuda.synchronize()
    e_dw = _rel(ops.weight_from_arena(dwd.cpu(), tr), w.grad)

    print(f'case={case} dtype={dtype} fwd={e_fwd:.3e} dx={e_dx:.3e} dw={e_dw:.3e}')
    assert e_fwd < tol, f'forward rel err {e_fwd}'
    assert e_dx < tol, f'dgrad rel err {e_dx}'
    assert e_dw < tol, f'wgrad rel err {e_dw}'


def test_conv_bias_and_identity(cuda):
    """A = I check with an asymmetric weight: catches transposed C/D maps."""
    from oct_segmentation_amd import ops
    Cin = Cout = 64
    x = torch.zeros(1, Cin, 8, 16)
    for c in range(Cin):
        x[0, c, c % 8, (3 * c) % 16] = 1.0 + c
    w = torch.arange(Cout * Cin, dtype=torch.float32).reshape(Cout, Cin, 1, 1) / 100.0
    b = torch.arange(Cout, dtype=torch.float32)
    y = F.conv2d(x, w, b)
    yd = ops.conv2d_forward(x.permute(0, 2, 3, 1).contiguous().to(cuda), ops.weight_to_arena(w).to(cuda),
                            b.to(cuda), 1, 0, False)
    torch.cuda.synchronize()
    assert _rel(yd.cpu().permute(0, 3, 1, 2), y) < 1e-6


# (CASES[17]: the persistent 3x3 kernel's head-of-tap variant, which the f16 instantiation uses)
@pytest.mark.parametrize('case', [CASES[0], CASES[3], CASES[4], CASES[6], CASES[7], CASES[12], CASES[13], CASES[17], CASES[20], CASES[21]], ids=str)
def test_conv_forward_f16(cuda, case):
    """IEEE half storage (OCTSEG_F16, the serving dtype of BASELINE config #5): forward of every conv family against torch on the
    same f16-quantised operands, f32 accumulate -> 2e-3 of the output scale; the backward entry points refuse the dtype."""
    from oct_segmentation_amd import ops
    N, H, W, Cin, Cout, R, stride, pad, tr = case
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(N, Cin, H, W, generator=g).half().float()
    wshape = (Cin, Cout, R, R) if tr else (Cout, Cin, R, R)
    w = (torch.randn(wshape, generator=g) / (Cin * R * R) ** 0.5).half().float()
    y = _ref(x, w, stride, pad, tr)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda, torch.float16)
    wa = ops.weight_to_arena(w, tr).to(cuda)
    yd = ops.conv2d_forward(xd, wa, None, stride, pad, tr)
    torch.cuda.synchronize()
    assert yd.dtype == torch.float16
    e = _rel(yd.float().cpu().permute(0, 3, 1, 2), y)
    print(f'f16 case={case} fwd={e:.3e}')
    assert e < 2e-3
    with pytest.raises(RuntimeError):
        ops.conv2d_backward_weight(xd, yd, R, stride, pad, tr)
