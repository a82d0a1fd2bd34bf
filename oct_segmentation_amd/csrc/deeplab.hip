// deeplab.hip -- the HBM-bound kernels that the DeepLabV3+ and PSPNet decoders add to the hot path (gfx950): depthwise (dilated) 3x3 convolution
// forward / data gradient / weight gradient, the space <-> batch permutation that turns ResNet layer4 at dilation 2 into ordinary 3x3
// convolutions, image pooling (global mean) and its broadcast, element-wise dropout with an injected keep mask, and the plain bilinear
// (align_corners=True) resample between NHWC tensors.
//
// Reference: `DeepLabV3Plus` is one of the architectures the reference sweeps (configs/tune.yaml:9-18 -> smp.create_model(arch=...),
// src/models/smp/model.py:38-44; DeepLabV3Plus/resnet101 is a per-class winner in eval/tuning/configs_best.xlsx); the arithmetic is
// smp 0.3.3 decoders/deeplabv3/{model,decoder}.py + base/modules.py SeparableConv2d + encoders/_base.py make_dilated, restated in
// oracle/nets.py (DeepLabV3PlusDecoder, ResNetEncoder.make_dilated).
//
// Dilation without a dilated kernel: a 3x3 conv with dilation 2 and padding 2 only ever combines pixels of equal row / column parity, and
// on each of the four parity sub-grids it IS a 3x3 conv with padding 1 (the zero border of the fine map is the zero border of every
// sub-grid).  make_dilated(16) sets stride 1 / dilation 2 on every conv of layer4, 1x1 convs and BatchNorm are position-blind, so layer4
// runs unchanged on the [4N][H/2][W/2] re-arrangement of layer3's output and its result is re-arranged back for the ASPP.
// All tensors NHWC, 16-byte vectors, f32 arithmetic; the only float atomics are the depthwise weight gradients (one per block and weight).
#include "common.h"
#include "ev.h"
#include "kernels.h"

namespace octseg {

#define DL_DISPATCH(KERNEL, grid, ...)                                                          \
  do {                                                                                          \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, st, __VA_ARGS__);      \
    else if (dtype == DT_F16) hipLaunchKernelGGL(KERNEL<f16_t>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)

template <typename T>
static __device__ __forceinline__ void put(void* dst, size_t idx, float* x, int accum) {
  constexpr int VEC = EV<T>::VEC;
  if (accum) {
    float o[VEC];
    EV<T>::unpack(ldv<T>(dst, idx), o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += o[i];
  }
  stv<T>(dst, idx, EV<T>::pack(x));
}

// ------------------------------------------------------------------ space <-> batch (dilation 2 as parity sub-grids)
// fine [N][H][W][C], coarse [4N][H/2][W/2][C], coarse image = 4 n + 2 (y & 1) + (x & 1).  to_coarse: coarse = fine; else fine (+)= coarse.
template <typename T>
__global__ __launch_bounds__(256) void parity_permute_kernel(const void* src, void* dst, int N, int H, int W, int vpc, int to_coarse, int accum) {
  constexpr int VEC = EV<T>::VEC;
  const size_t nvec = (size_t)N * H * W * vpc;
  const int H2 = H >> 1, W2 = W >> 1;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    const size_t c = ((((size_t)n * 4 + (y & 1) * 2 + (x & 1)) * H2 + (y >> 1)) * W2 + (x >> 1)) * vpc + cv;
    float f[VEC];
    EV<T>::unpack(ldv<T>(src, to_coarse ? v : c), f);
    put<T>(dst, to_coarse ? c : v, f, accum);
  }
}
hipError_t launch_parity_permute(int dtype, const void* src, void* dst, int N, int H, int W, int C, int to_coarse, int accum, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * H * W * (C / vec);
  DL_DISPATCH(parity_permute_kernel, dim3(grid_for(nvec, 256)), src, dst, N, H, W, C / vec, to_coarse, accum);
  return hipGetLastError();
}

// ------------------------------------------------------------------ depthwise 3x3 (dilation d, padding d, stride 1)
// out[n][y][x][oc0 + c] (+)= sum_t w[t'][wc0 + c] * in[n][y + d (r - 1)][x + d (s - 1)][ic0 + c],  t = 3 r + s, t' = flip ? 8 - t : t
// (flip = the data gradient: correlation with the mirrored kernel).  Weights: the fp32 master [3][3][Cw] (tap-major, channel-contiguous).
// `in` / `out` may be channel slices of wider tensors (inC / outC = their channel counts): torch.cat never materialises.
struct DwArgs {
  const void* in; void* out; const float* w;
  int inC, ic0, outC, oc0, wC, wc0;
  int N, H, W, C, dil, flip, accum;
};
template <typename T>
__global__ __launch_bounds__(256) void dw_conv_kernel(const DwArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC, ivs = a.inC / VEC, ovs = a.outC / VEC, iv0 = a.ic0 / VEC, ov0 = a.oc0 / VEC;
  const size_t nvec = (size_t)a.N * a.H * a.W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int x = (int)(p % a.W); p /= a.W;
    const int y = (int)(p % a.H);
    const int n = (int)(p / a.H);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + a.dil * (t / 3 - 1), xx = x + a.dil * (t % 3 - 1);
      if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
      float f[VEC];
      EV<T>::unpack(ldv<T>(a.in, (((size_t)n * a.H + yy) * a.W + xx) * ivs + iv0 + cv), f);
      const float* wt = a.w + (size_t)(a.flip ? 8 - t : t) * a.wC + a.wc0 + cv * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wt[i], f[i], acc[i]);
    }
    put<T>(a.out, (((size_t)n * a.H + y) * a.W + x) * ovs + ov0 + cv, acc, a.accum);
  }
}
static bool dw_ok(int dtype, const DwArgs& a) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  return a.C % vec == 0 && a.inC % vec == 0 && a.outC % vec == 0 && a.ic0 % vec == 0 && a.oc0 % vec == 0 && a.wc0 % 4 == 0 && a.wC % 4 == 0 &&
         a.dil >= 1 && a.ic0 + a.C <= a.inC && a.oc0 + a.C <= a.outC && a.wc0 + a.C <= a.wC;
}
hipError_t launch_dw_conv(int dtype, const void* in, int inC, int ic0, void* out, int outC, int oc0, const float* w, int wC, int wc0, int N, int H,
                          int W, int C, int dil, int flip, int accum, hipStream_t st) {
  DwArgs a{in, out, w, inC, ic0, outC, oc0, wC, wc0, N, H, W, C, dil, flip, accum};
  if (!dw_ok(dtype, a)) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * H * W * (C / (dtype == DT_F32 ? 4 : 8));
  DL_DISPATCH(dw_conv_kernel, dim3(grid_for(nvec, 256)), a);
  return hipGetLastError();
}

// weight gradient: dw[t][wc0 + c] += sum_{n, y, x} gout[n][y][x][oc0 + c] * in[n][y + d (r - 1)][x + d (s - 1)][ic0 + c].
// Block (bx, by): channel vectors [DW_CH by, DW_CH by + nv), thread (r, v) owns vector v of the pixels r, r + rows, ... of its share; nine
// taps x VEC sums in registers, folded over the rows through LDS one tap at a time, one atomic per block and weight.  The atomics
// number (pixel groups) x 9 x C whatever the channel split, so the grid gets its width from 32-vector channel chunks (>= 8 pixel rows
// per block) and about 1024 blocks in all (first version: 256-vector chunks x 256 pixel groups, one pixel row per block: 0.5 ms per launch).
constexpr int DW_CH = 32;   // channel vectors per block of the reductions below
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwArgs a, float* dw) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256 * VEC];
  const int vpc = a.C / VEC, ivs = a.inC / VEC, ovs = a.outC / VEC, iv0 = a.ic0 / VEC, ov0 = a.oc0 / VEC;
  const int v0 = blockIdx.y * DW_CH, nv = min(DW_CH, vpc - v0), rows = 256 / nv;
  const int r = threadIdx.x / nv, cv = threadIdx.x - r * nv;
  float acc[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
  const size_t npix = (size_t)a.N * a.H * a.W;
  if (r < rows)
    for (size_t p = (size_t)blockIdx.x * rows + r; p < npix; p += (size_t)gridDim.x * rows) {
      const int x = (int)(p % a.W);
      const int y = (int)((p / a.W) % a.H);
      const size_t n = p / ((size_t)a.W * a.H);
      float g[VEC];
      EV<T>::unpack(ldv<T>(a.out, p * ovs + ov0 + v0 + cv), g);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + a.dil * (t / 3 - 1), xx = x + a.dil * (t % 3 - 1);
        if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
        float f[VEC];
        EV<T>::unpack(ldv<T>(a.in, ((n * a.H + yy) * a.W + xx) * ivs + iv0 + v0 + cv), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[t][i] = fmaf(g[i], f[i], acc[t][i]);
      }
    }
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (r < rows)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[(r * nv + cv) * VEC + i] = acc[t][i];
    __syncthreads();
    for (int c = threadIdx.x; c < nv * VEC; c += 256) {
      float s = 0.f;
      for (int k = 0; k < rows; ++k) s += red[k * nv * VEC + c];
      atomicAdd(dw + (size_t)t * a.wC + a.wc0 + v0 * VEC + c, s);
    }
  }
}
hipError_t launch_dw_wgrad(int dtype, const void* in, int inC, int ic0, const void* gout, int goC, int oc0, float* dw, int wC, int wc0, int N, int H,
                           int W, int C, int dil, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  DwArgs a{in, const_cast<void*>(gout), nullptr, inC, ic0, goC, oc0, wC, wc0, N, H, W, C, dil, 0, 0};
  if (!dw_ok(dtype, a)) return hipErrorInvalidValue;
  const int vpc = C / (dtype == DT_F32 ? 4 : 8);
  const int nv = vpc < DW_CH ? vpc : DW_CH, rows = 256 / nv;
  const size_t npix = (size_t)N * H * W;
  const int nch = (vpc + DW_CH - 1) / DW_CH;
  size_t gx = (npix + (size_t)rows * 8 - 1) / ((size_t)rows * 8);      // >= 8 pixels per thread ...
  const size_t want = (size_t)(1024 + nch - 1) / nch;                   // ... and about 1024 blocks in all (four per CU)
  if (gx > want) gx = want;
  if (gx < 1 || deterministic_mode()) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)nch);
  if (dtype == DT_F32) hipLaunchKernelGGL(dw_wgrad_kernel<float>, grid, dim3(256), 0, st, a, dw);
  else hipLaunchKernelGGL(dw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, a, dw);
  return hipGetLastError();
}

// ------------------------------------------------------------------ per-image channel sums and their broadcast
// out[n][c] = sum_p in[n][p][c] / div   (AdaptiveAvgPool2d(1): div = HW; gradient of the broadcast below: div = 1).
// grid (ceil(vpc / DW_CH), N): deterministic (one block per image and 32-vector channel chunk, >= 8 pixel rows each, fixed order).
template <typename T>
__global__ __launch_bounds__(256) void image_sum_kernel(const void* in, void* out, int HW, int vpc, float div) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256 * VEC];
  const int v0 = blockIdx.x * DW_CH, nv = min(DW_CH, vpc - v0), rows = 256 / nv;
  const int r = threadIdx.x / nv, cv = threadIdx.x - r * nv;
  const size_t n = blockIdx.y;
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  if (r < rows)
    for (int p = r; p < HW; p += rows) {
      float f[VEC];
      EV<T>::unpack(ldv<T>(in, (n * HW + p) * vpc + v0 + cv), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += f[i];
    }
  if (r < rows)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(r * nv + cv) * VEC + i] = s[i];
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rows; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += red[(k * nv + cv) * VEC + i];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] /= div;   // (a division, as torch's mean)
    stv<T>(out, n * vpc + v0 + cv, EV<T>::pack(s));
  }
}
hipError_t launch_image_sum(int dtype, const void* in, void* out, int N, int HW, int C, float div, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const int vpc = C / vec;
  DL_DISPATCH(image_sum_kernel, dim3((vpc + DW_CH - 1) / DW_CH, N), in, out, HW, vpc, div);
  return hipGetLastError();
}
// out[n][p][c] (+)= scale * in[n][c]   (F.interpolate of a 1x1 map to any size, either align_corners: a broadcast; gradient of the mean)
template <typename T>
__global__ __launch_bounds__(256) void image_bcast_kernel(const void* in, void* out, size_t HW, int vpc, float scale, int accum, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    const size_t n = v / (HW * vpc);
    float f[VEC];
    EV<T>::unpack(ldv<T>(in, n * vpc + cv), f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] *= scale;
    put<T>(out, v, f, accum);
  }
}
hipError_t launch_image_bcast(int dtype, const void* in, void* out, int N, int HW, int C, float scale, int accum, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * HW * (C / vec);
  DL_DISPATCH(image_bcast_kernel, dim3(grid_for(nvec, 256)), in, out, (size_t)HW, C / vec, scale, accum, nvec);
  return hipGetLastError();
}

// ------------------------------------------------------------------ element-wise dropout with an injected keep mask (nn.Dropout(0.5) of ASPP.project)
// out = in * keep * mscale;  keep: float 0 / 1, NHWC like the tensor (nullptr: copy).  The gradient is the same kernel on the gradient.
template <typename T>
__global__ __launch_bounds__(256) void drop_elem_kernel(const void* in, const float* keep, float mscale, void* out, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float f[VEC];
    EV<T>::unpack(ldv<T>(in, v), f);
    if (keep != nullptr) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] *= keep[v * VEC + i] * mscale;
    }
    stv<T>(out, v, EV<T>::pack(f));
  }
}
hipError_t launch_drop_elem(int dtype, const void* in, const float* keep, float mscale, void* out, size_t numel, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (numel % vec != 0) return hipErrorInvalidValue;
  const size_t nvec = numel / vec;
  DL_DISPATCH(drop_elem_kernel, dim3(grid_for(nvec, 256)), in, keep, mscale, out, nvec);
  return hipGetLastError();
}

// ------------------------------------------------------------------ bilinear resample by `up`, align_corners=True (nn.UpsamplingBilinear2d), NHWC -> NHWC
// torch's source index: scale = (in - 1) / (out - 1) in float, x = scale * o, i0 = (int)x, lambda1 = x - i0 (as fpn.hip; the adjoint there
// uses the same expressions, launch_bilinear_adjoint is this kernel's backward)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up_kernel(const void* in, void* out, int N, int H, int W, int vpc, int up, float sy, float sx) {
  constexpr int VEC = EV<T>::VEC;
  const int OH = H * up, OW = W * up;
  const size_t nvec = (size_t)N * OH * OW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const size_t n = p / OH;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    float a00[VEC], a01[VEC], a10[VEC], a11[VEC], o[VEC];
    EV<T>::unpack(ldv<T>(in, ((n * H + y0) * W + x0) * vpc + cv), a00);
    EV<T>::unpack(ldv<T>(in, ((n * H + y0) * W + x1) * vpc + cv), a01);
    EV<T>::unpack(ldv<T>(in, ((n * H + y1) * W + x0) * vpc + cv), a10);
    EV<T>::unpack(ldv<T>(in, ((n * H + y1) * W + x1) * vpc + cv), a11);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = wy0 * (wx0 * a00[i] + wx1 * a01[i]) + wy1 * (wx0 * a10[i] + wx1 * a11[i]);
    stv<T>(out, v, EV<T>::pack(o));
  }
}
hipError_t launch_bilinear_up(int dtype, const void* in, void* out, int N, int H, int W, int C, int up, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || up < 2) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * H * up * W * up * (C / vec);
  const float sy = H * up > 1 ? (float)(H - 1) / (float)(H * up - 1) : 0.f, sx = W * up > 1 ? (float)(W - 1) / (float)(W * up - 1) : 0.f;
  DL_DISPATCH(bilinear_up_kernel, dim3(grid_for(nvec, 256)), in, out, N, H, W, C / vec, up, sy, sx);
  return hipGetLastError();
}

// ================================================================== PSPNet (smp decoders/pspnet, restated in oracle/nets.py PSPDecoder)
// ------------------------------------------------------------------ nn.AdaptiveAvgPool2d((k, k)): bin i covers [floor(i H / k), ceil((i + 1) H / k))
// (neighbouring bins overlap by a row / column when k does not divide H).  out [N][k][k][C]; grid (ceil(vpc / DW_CH), N k k), deterministic.
static __device__ __forceinline__ void bin_range(int i, int k, int H, int& lo, int& hi) { lo = (i * H) / k; hi = ((i + 1) * H + k - 1) / k; }
template <typename T>
__global__ __launch_bounds__(256) void bin_mean_kernel(const void* in, void* out, int H, int W, int k, int vpc, int ch) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256 * VEC];
  const int v0 = blockIdx.x * ch, nv = min(ch, vpc - v0), rows = 256 / nv;   // ch channel vectors per block (few bins: narrow chunks, more pixel rows)
  const int r = threadIdx.x / nv, cv = threadIdx.x - r * nv;
  const int bin = blockIdx.y % (k * k);
  const size_t n = blockIdx.y / (k * k);
  int y0, y1, x0, x1;
  bin_range(bin / k, k, H, y0, y1);
  bin_range(bin % k, k, W, x0, x1);
  const int bw = x1 - x0, cnt = (y1 - y0) * bw;
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  if (r < rows)
    for (int p = r; p < cnt; p += rows) {
      const int y = y0 + p / bw, x = x0 + p % bw;
      float f[VEC];
      EV<T>::unpack(ldv<T>(in, ((n * H + y) * W + x) * vpc + v0 + cv), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += f[i];
    }
  if (r < rows)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(r * nv + cv) * VEC + i] = s[i];
  __syncthreads();
  if (r == 0) {
    for (int q = 1; q < rows; ++q)
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += red[(q * nv + cv) * VEC + i];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] /= (float)cnt;
    stv<T>(out, (size_t)blockIdx.y * vpc + v0 + cv, EV<T>::pack(s));
  }
}
hipError_t launch_bin_mean(int dtype, const void* in, void* out, int N, int H, int W, int C, int k, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || k < 1) return hipErrorInvalidValue;
  const int vpc = C / vec;
  const int ch = N * k * k >= 256 ? DW_CH : 8;      // the 1x1 / 2x2 bins of a 16-frame batch are 16 / 64 blocks of up to 7744 pixels each
  DL_DISPATCH(bin_mean_kernel, dim3((vpc + ch - 1) / ch, N * k * k), in, out, H, W, k, vpc, ch);
  return hipGetLastError();
}
// its gradient, gather form: gin[n][y][x] (+)= sum over the bins that contain (y, x) of gout[bin] / area(bin)
template <typename T>
__global__ __launch_bounds__(256) void bin_mean_bwd_kernel(const void* gout, void* gin, int N, int H, int W, int k, int vpc, int accum) {
  constexpr int VEC = EV<T>::VEC;
  const size_t nvec = (size_t)N * H * W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const size_t n = p / H;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int bi = 0; bi < k; ++bi) {            // (k <= 6: every bin is tried; with k > H several bins hold the same pixel)
      int y0, y1;
      bin_range(bi, k, H, y0, y1);
      if (y < y0 || y >= y1) continue;
      for (int bj = 0; bj < k; ++bj) {
        int x0, x1;
        bin_range(bj, k, W, x0, x1);
        if (x < x0 || x >= x1) continue;
        float f[VEC];
        EV<T>::unpack(ldv<T>(gout, ((n * k + bi) * k + bj) * vpc + cv), f);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(f[i], inv, acc[i]);
      }
    }
    put<T>(gin, v, acc, accum);
  }
}
hipError_t launch_bin_mean_bwd(int dtype, const void* gout, void* gin, int N, int H, int W, int C, int k, int accum, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || k < 1) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * H * W * (C / vec);
  if (dtype == DT_F32) hipLaunchKernelGGL(bin_mean_bwd_kernel<float>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, gout, gin, N, H, W, k, C / vec, accum);
  else hipLaunchKernelGGL(bin_mean_bwd_kernel<bf16_t>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, gout, gin, N, H, W, k, C / vec, accum);
  return hipGetLastError();
}

// ------------------------------------------------------------------ F.interpolate(size=(OH, OW), mode='bilinear', align_corners=True), any size pair
struct Tap2 { int i0, i1; float w0, w1; };
static __device__ __forceinline__ Tap2 tap_of(int o, int in, float scale) {   // torch: x = scale * o, i0 = (int)x, lambda1 = x - i0
  const float x = scale * (float)o;
  Tap2 t;
  t.i0 = min((int)x, in - 1);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.w1 = x - (float)t.i0;
  t.w0 = 1.f - t.w1;
  return t;
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_resize_kernel(const void* in, void* out, int N, int IH, int IW, int OH, int OW, int vpc, float sy, float sx) {
  constexpr int VEC = EV<T>::VEC;
  const size_t nvec = (size_t)N * OH * OW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const size_t n = p / OH;
    const Tap2 ty = tap_of(oy, IH, sy), tx = tap_of(ox, IW, sx);
    float a00[VEC], a01[VEC], a10[VEC], a11[VEC], o[VEC];
    EV<T>::unpack(ldv<T>(in, ((n * IH + ty.i0) * IW + tx.i0) * vpc + cv), a00);
    EV<T>::unpack(ldv<T>(in, ((n * IH + ty.i0) * IW + tx.i1) * vpc + cv), a01);
    EV<T>::unpack(ldv<T>(in, ((n * IH + ty.i1) * IW + tx.i0) * vpc + cv), a10);
    EV<T>::unpack(ldv<T>(in, ((n * IH + ty.i1) * IW + tx.i1) * vpc + cv), a11);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = ty.w0 * (tx.w0 * a00[i] + tx.w1 * a01[i]) + ty.w1 * (tx.w0 * a10[i] + tx.w1 * a11[i]);
    stv<T>(out, v, EV<T>::pack(o));
  }
}
static inline float resize_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
hipError_t launch_bilinear_resize(int dtype, const void* in, void* out, int N, int IH, int IW, int OH, int OW, int C, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * OH * OW * (C / vec);
  DL_DISPATCH(bilinear_resize_kernel, dim3(grid_for(nvec, 256)), in, out, N, IH, IW, OH, OW, C / vec, resize_scale(IH, OH), resize_scale(IW, OW));
  return hipGetLastError();
}
// adjoint, gather form (the source is a handful of pixels: k x k bins): one block per (image, source pixel, 32-vector channel chunk) walks
// the outputs that can touch that pixel -- its threads share them out, fold through LDS in a fixed order -- and adds the very weights the
// forward used; one writer per element.  (One THREAD per source element took 12 ms of a 25 ms PSPNet step: 256 threads walking 88 x 88
// outputs each for the 1x1 bin.)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_resize_adjoint_kernel(const void* gout, void* gin, int IH, int IW, int OH, int OW, int vpc,
                                                                      float sy, float sx) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256 * VEC];
  const int v0 = blockIdx.x * DW_CH, nv = min(DW_CH, vpc - v0), rows = 256 / nv;
  const int r = threadIdx.x / nv, cv = threadIdx.x - r * nv;
  const size_t bidx = (size_t)blockIdx.z * gridDim.y + blockIdx.y;   // (source pixels beyond 65535 continue in grid.z: PAN's GAU maps)
  const int pix = (int)(bidx % (size_t)(IH * IW));
  const size_t n = bidx / (size_t)(IH * IW);
  const int iy = pix / IW, ix = pix - iy * IW;
  int oy0 = 0, oy1 = OH - 1, ox0 = 0, ox1 = OW - 1;
  if (sy > 0.f) { oy0 = max(0, (int)floorf((float)(iy - 1) / sy) - 1); oy1 = min(OH - 1, (int)ceilf((float)(iy + 1) / sy) + 1); }
  if (sx > 0.f) { ox0 = max(0, (int)floorf((float)(ix - 1) / sx) - 1); ox1 = min(OW - 1, (int)ceilf((float)(ix + 1) / sx) + 1); }
  const int bw = ox1 - ox0 + 1, cnt = (oy1 - oy0 + 1) * bw;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (r < rows)
    for (int q = r; q < cnt; q += rows) {
      const int oy = oy0 + q / bw, ox = ox0 + q % bw;
      const Tap2 ty = tap_of(oy, IH, sy), tx = tap_of(ox, IW, sx);
      const float w = ((ty.i0 == iy ? ty.w0 : 0.f) + (ty.i1 == iy ? ty.w1 : 0.f)) * ((tx.i0 == ix ? tx.w0 : 0.f) + (tx.i1 == ix ? tx.w1 : 0.f));
      if (w == 0.f) continue;
      float f[VEC];
      EV<T>::unpack(ldv<T>(gout, ((n * OH + oy) * OW + ox) * vpc + v0 + cv), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(f[i], w, acc[i]);
    }
  if (r < rows)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(r * nv + cv) * VEC + i] = acc[i];
  __syncthreads();
  if (r == 0) {
    for (int q = 1; q < rows; ++q)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += red[(q * nv + cv) * VEC + i];
    stv<T>(gin, bidx * vpc + v0 + cv, EV<T>::pack(acc));
  }
}
hipError_t launch_bilinear_resize_adjoint(int dtype, const void* gout, void* gin, int N, int IH, int IW, int OH, int OW, int C, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const int vpc = C / vec;
  const float sy = resize_scale(IH, OH), sx = resize_scale(IW, OW);
  const long long npx = (long long)N * IH * IW;
  int gy = (int)npx, gz = 1;
  if (npx > 65535) {                       // one workgroup per source pixel: factor the count into grid.y x grid.z exactly
    gy = 0;
    for (int d = 65535; d >= 1; --d) if (npx % d == 0) { gy = d; break; }
    gz = (int)(npx / gy);
    if (gz > 65535) return hipErrorInvalidValue;
  }
  const dim3 grid((vpc + DW_CH - 1) / DW_CH, gy, gz);
  if (dtype == DT_F32) hipLaunchKernelGGL(bilinear_resize_adjoint_kernel<float>, grid, dim3(256), 0, st, gout, gin, IH, IW, OH, OW, vpc, sy, sx);
  else hipLaunchKernelGGL(bilinear_resize_adjoint_kernel<bf16_t>, grid, dim3(256), 0, st, gout, gin, IH, IW, OH, OW, vpc, sy, sx);
  return hipGetLastError();
}

// ------------------------------------------------------------------ ReLU of a plain tensor (PSPBlock with pool_size 1: biased conv, no BatchNorm), and its gradient
// fwd: out = max(in, 0) (NaN propagates);  bwd (mask = the forward's OUTPUT): gin = gout where out > 0
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(const void* in, const void* mask, void* out, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float f[VEC], m[VEC];
    EV<T>::unpack(ldv<T>(in, v), f);
    if (mask != nullptr) {
      EV<T>::unpack(ldv<T>(mask, v), m);
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] = m[i] > 0.f ? f[i] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] = f[i] < 0.f ? 0.f : f[i];
    }
    stv<T>(out, v, EV<T>::pack(f));
  }
}
hipError_t launch_relu(int dtype, const void* in, const void* mask, void* out, size_t numel, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (numel % vec != 0) return hipErrorInvalidValue;
  DL_DISPATCH(relu_kernel, dim3(grid_for(numel / vec, 256)), in, mask, out, numel / vec);
  return hipGetLastError();
}

// ================================================================== DeepLabV3 (dense ASPP): dilation r, any rate, without a dilated kernel
// A 3x3 conv with dilation r and padding r combines pixels of equal (y mod r, x mod r) only: on each of the r^2 sub-grids it is the plain
// pad-1 3x3 conv.  For r = 12 / 24 / 36 the sub-grids of an 88 x 88 map are 8 x 8 .. 3 x 3 pixels -- far below a conv tile -- so they are
// laid out as ONE mosaic image per frame instead of as a batch: sub-grid (a, b) occupies an hs x ws block at rows 1 + a (hs + 1), columns
// 1 + b (ws + 1) (hs = ceil(H / r)), separated by one-pixel ZERO gutters, which are exactly the zero padding each sub-grid's conv needs.
// A plain 3x3 / pad 1 conv over the (r (hs + 1) + 1)-square mosaic then computes every sub-grid at once; outputs on gutters and on the
// padding behind H, W are discarded on the way back.  (BatchNorm statistics therefore come from tensor_stats on the fine tensor, not
// from the conv's epilogue.)
// to_mosaic: mosaic = fine re-arranged, zeros elsewhere (every mosaic element is written);  else: fine (+)= its mosaic element.
template <typename T>
__global__ __launch_bounds__(256) void mosaic_kernel(const void* src, void* dst, int N, int H, int W, int vpc, int r, int hs, int ws, int to_mosaic,
                                                     int accum) {
  constexpr int VEC = EV<T>::VEC;
  const int MH = r * (hs + 1) + 1, MW = r * (ws + 1) + 1;
  if (to_mosaic) {
    const size_t nvec = (size_t)N * MH * MW * vpc;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
      const int cv = (int)(v % vpc);
      size_t p = v / vpc;
      const int mx = (int)(p % MW); p /= MW;
      const int my = (int)(p % MH);
      const size_t n = p / MH;
      uint4 q = make_uint4(0, 0, 0, 0);
      if (my >= 1 && mx >= 1) {
        const int a = (my - 1) / (hs + 1), i = (my - 1) - a * (hs + 1);
        const int b = (mx - 1) / (ws + 1), j = (mx - 1) - b * (ws + 1);
        const int y = i * r + a, x = j * r + b;
        if (i < hs && j < ws && a < r && b < r && y < H && x < W) q = ldv<T>(src, ((n * H + y) * W + x) * vpc + cv);
      }
      stv<T>(dst, v, q);
    }
  } else {
    const size_t nvec = (size_t)N * H * W * vpc;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
      const int cv = (int)(v % vpc);
      size_t p = v / vpc;
      const int x = (int)(p % W); p /= W;
      const int y = (int)(p % H);
      const size_t n = p / H;
      const int my = 1 + (y % r) * (hs + 1) + y / r, mx = 1 + (x % r) * (ws + 1) + x / r;
      float f[VEC];
      EV<T>::unpack(ldv<T>(src, ((n * MH + my) * MW + mx) * vpc + cv), f);
      put<T>(dst, v, f, accum);
    }
  }
}
hipError_t launch_mosaic(int dtype, const void* src, void* dst, int N, int H, int W, int C, int r, int to_mosaic, int accum, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || r < 1) return hipErrorInvalidValue;
  const int hs = (H + r - 1) / r, ws = (W + r - 1) / r;
  const size_t nvec = to_mosaic ? (size_t)N * (r * (hs + 1) + 1) * (r * (ws + 1) + 1) * (C / vec) : (size_t)N * H * W * (C / vec);
  DL_DISPATCH(mosaic_kernel, dim3(grid_for(nvec, 256)), src, dst, N, H, W, C / vec, r, hs, ws, to_mosaic, accum);
  return hipGetLastError();
}

// BatchNorm partial sums of a plain NHWC tensor: slab[row][c] = (sum y, sum y^2) over the pixels row, row + rows, ... (the rows of
// bn_finalize_train; for tensors whose producer is not a conv epilogue).  C / VEC <= 256 channel vectors.
template <typename T>
__global__ __launch_bounds__(256) void tensor_stats_kernel(const void* y, size_t npix, int vpc, int C, float* slab, int vstride, int cv0) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256][2 * VEC + 1];
  const int tpv = 256 / vpc;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (pl < tpv)
    for (size_t p = (size_t)blockIdx.x * tpv + pl; p < npix; p += (size_t)gridDim.x * tpv) {
      float f[VEC];
      EV<T>::unpack(ldv<T>(y, p * vstride + cv0 + cv), f);     // (a chunk of vpc <= 256 channel vectors of a tensor with vstride per pixel)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s1[i] += f[i]; s2[i] = fmaf(f[i], f[i], s2[i]); }
    }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[threadIdx.x][i] = s1[i]; red[threadIdx.x][VEC + i] = s2[i]; }
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < tpv; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s1[i] += red[threadIdx.x + k * vpc][i]; s2[i] += red[threadIdx.x + k * vpc][VEC + i]; }
    float* o = slab + ((size_t)blockIdx.x * C + (cv0 + cv) * VEC) * 2;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { o[2 * i] = s1[i]; o[2 * i + 1] = s2[i]; }
  }
}
hipError_t launch_tensor_stats(int dtype, const void* y, size_t npix, int C, float* slab, int rows, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || rows < 1) return hipErrorInvalidValue;
  const int vtot = C / vec;
  for (int cv0 = 0; cv0 < vtot; cv0 += 256) {      // wide tensors (RegNet stage 4: 1624 / 2240 channels) in chunks of 256 channel vectors
    const int vpc = vtot - cv0 < 256 ? vtot - cv0 : 256;
    if (dtype == DT_F32) hipLaunchKernelGGL(tensor_stats_kernel<float>, dim3(rows), dim3(256), 0, st, y, npix, vpc, C, slab, vtot, cv0);
    else hipLaunchKernelGGL(tensor_stats_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, y, npix, vpc, C, slab, vtot, cv0);
  }
  return hipGetLastError();
}

}  // namespace octseg
