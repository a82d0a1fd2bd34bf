"""Single-op bindings (NHWC tensors on the GPU) over the C ABI.

They exist so that every HIP kernel family can be pinned against torch CPU in
isolation (tests/test_gpu_ops.py); the network path uses the plan API in
``engine.py`` instead.
"""
import torch

from . import _lib as L


def _dt(t):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float16:
        return L.F16   # forward only
    raise TypeError(f'unsupported dtype {t.dtype}: float32, bfloat16 or float16 expected')


def weight_to_arena(w, transposed=False):
    """torch Conv2d [O,I,R,S] (or ConvTranspose2d [I,O,R,S]) -> arena layout [R,S,O,I] fp32."""
    w = w.detach().float()
    return (w.permute(2, 3, 1, 0) if transposed else w.permute(2, 3, 0, 1)).contiguous()


def weight_from_arena(wa, transposed=False):
    return (wa.permute(3, 2, 0, 1) if transposed else wa.permute(2, 3, 0, 1)).contiguous()


def _out_hw(H, W, R, stride, pad, transposed):
    if transposed:
        return H * 2, W * 2
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1


def _scratch(dt, N, H, W, Cin, Cout, R, dev):
    n = L.lib().octseg_conv2d_scratch_bytes(dt, N, H, W, Cin, Cout, R, R)
    return torch.empty(n, dtype=torch.uint8, device=dev)


def conv2d_forward(x, w_arena, bias=None, stride=1, pad=0, transposed=False):
    """x: [N,H,W,Cin] NHWC contiguous (f32|bf16, cuda); w_arena: [R,S,O,I] f32 cuda."""
    assert x.is_cuda and x.is_contiguous() and w_arena.is_contiguous() and w_arena.dtype == torch.float32
    N, H, W, Cin = x.shape
    R, S, O, I = w_arena.shape
    assert I == Cin
    OH, OW = _out_hw(H, W, R, stride, pad, transposed)
    y = torch.empty((N, OH, OW, O), dtype=x.dtype, device=x.device)
    dt = _dt(x)
    sc = _scratch(dt, N, H, W, Cin, O, R, x.device)
    L.check(L.lib().octseg_conv2d_forward(dt, L.ptr(x), L.ptr(w_arena), L.ptr(bias), L.ptr(y), N, H, W, Cin, O, R, S,
                                          stride, pad, int(transposed), L.ptr(sc), L.stream_ptr()))
    return y


def conv2d_backward_data(dy, w_arena, in_hw, stride=1, pad=0, transposed=False):
    assert dy.is_cuda and dy.is_contiguous()
    N, OH, OW, O = dy.shape
    R, S, O2, I = w_arena.shape
    assert O2 == O
    H, W = in_hw
    dx = torch.empty((N, H, W, I), dtype=dy.dtype, device=dy.device)
    dt = _dt(dy)
    sc = _scratch(dt, N, H, W, I, O, R, dy.device)
    L.check(L.lib().octseg_conv2d_backward_data(dt, L.ptr(dy), L.ptr(w_arena), L.ptr(dx), N, H, W, I, O, R, S, stride,
                                                pad, int(transposed), L.ptr(sc), L.stream_ptr()))
    return dx


def conv2d_backward_weight(x, dy, R, stride=1, pad=0, transposed=False):
    assert x.is_cuda and x.is_contiguous() and dy.is_contiguous()
    N, H, W, I = x.shape
    O = dy.shape[3]
    dw = torch.empty((R, R, O, I), dtype=torch.float32, device=x.device)
    L.check(L.lib().octseg_conv2d_backward_weight(_dt(x), L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, W, I, O, R, R, stride,
                                                  pad, int(transposed), L.stream_ptr()))
    return dw
