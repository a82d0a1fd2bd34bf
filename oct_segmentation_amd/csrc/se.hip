// se.hip -- the squeeze-excite gate (timm SEModule inside RegNetY's bottleneck, smp 'timm-regnety_120'; reference sweep
// configs/tune.yaml:24 through smp.create_model, src/models/smp/model.py:38-44):
//
//     s = fc2(relu(fc1(mean_hw(x))))            -- mean: launch_image_sum, the two FCs: 1x1 convs on [N][1][1][C] maps (conv kernels)
//     out[n][p][c] = x[n][p][c] * sigmoid(s[n][c])
//
// Forward and the x-side gradient are the same HBM-bound sweep (dx (+)= g * sigmoid(s)); the s-side gradient
//     ds[n][c] = sigmoid'(s[n][c]) * sum_p g[n][p][c] * x[n][p][c]
// is a per-image reduction: one workgroup per (image, chunk of 32 channel vectors, pixel share), partial sums in a float scratch,
// finished by a second tiny launch -- fixed summation order, no atomics.
#include "common.h"
#include "ev.h"
#include "kernels.h"

namespace octseg {

#define SE_DISPATCH(KERNEL, grid, ...)                                                          \
  do {                                                                                          \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, st, __VA_ARGS__);      \
    else if (dtype == DT_F16) hipLaunchKernelGGL(KERNEL<f16_t>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)

static __device__ __forceinline__ float se_sigmoid(float z) {
  const float e = expf(-fabsf(z));                    // (no overflow for large |z|)
  return z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

// out (+)= in * sigmoid(s[n][c])
template <typename T>
__global__ __launch_bounds__(256) void se_gate_kernel(const void* in, const void* s, void* out, size_t HW, int vpc, int accum, size_t nvec, const void* s2) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    const size_t n = v / (HW * vpc);
    float f[VEC], g[VEC];
    EV<T>::unpack(ldv<T>(in, v), f);
    EV<T>::unpack(ldv<T>(s, n * vpc + cv), g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = se_sigmoid(g[i]);
    if (s2 != nullptr) {       // MAnet's MFAB: attention_hl + attention_ll, each behind its own sigmoid
      float g2[VEC];
      EV<T>::unpack(ldv<T>(s2, n * vpc + cv), g2);
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] += se_sigmoid(g2[i]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] *= g[i];
    if (accum) {
      float o[VEC];
      EV<T>::unpack(ldv<T>(out, v), o);
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] += o[i];
    }
    stv<T>(out, v, EV<T>::pack(f));
  }
}
hipError_t launch_se_gate(int dtype, const void* in, const void* s, void* out, int N, int HW, int C, int accum, hipStream_t st, const void* s2) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * HW * (C / vec);
  SE_DISPATCH(se_gate_kernel, dim3(grid_for(nvec, 256)), in, s, out, (size_t)HW, C / vec, accum, nvec, s2);
  return hipGetLastError();
}

constexpr int SE_CH = 32;     // channel vectors per workgroup
// part[n][share][c] = sum over the pixels share, share + nshare, ... of g * x      grid (chunks, shares, N)
template <typename T>
__global__ __launch_bounds__(256) void se_dgate_part_kernel(const void* g, const void* x, float* part, int HW, int vpc) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256 * VEC];
  const int v0 = blockIdx.x * SE_CH, nv = min(SE_CH, vpc - v0), rows = 256 / nv;
  const int r = threadIdx.x / nv, cv = threadIdx.x - r * nv;
  const size_t n = blockIdx.z;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (r < rows)
    for (int p = blockIdx.y * rows + r; p < HW; p += gridDim.y * rows) {
      float a[VEC], b[VEC];
      const size_t idx = (n * HW + p) * vpc + v0 + cv;
      EV<T>::unpack(ldv<T>(g, idx), a);
      EV<T>::unpack(ldv<T>(x, idx), b);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a[i], b[i], acc[i]);
    }
  if (r < rows)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(r * nv + cv) * VEC + i] = acc[i];
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rows; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += red[(k * nv + cv) * VEC + i];
    float* o = part + ((n * gridDim.y + blockIdx.y) * (size_t)vpc + v0 + cv) * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = acc[i];
  }
}
// ds[n][c] = sigmoid'(s[n][c]) * sum over the shares (in share order)
template <typename T>
__global__ __launch_bounds__(256) void se_dgate_fin_kernel(const float* part, const void* s, void* ds, int shares, int vpc, int nvec, const void* s2, void* ds2) {
  constexpr int VEC = EV<T>::VEC;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const int n = v / vpc, cv = v - n * vpc;
    float acc[VEC], z[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int k = 0; k < shares; ++k) {
      const float* p = part + (((size_t)n * shares + k) * vpc + cv) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += p[i];
    }
    EV<T>::unpack(ldv<T>(s, v), z);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { const float q = se_sigmoid(z[i]); o[i] = acc[i] * q * (1.0f - q); }
    stv<T>(ds, v, EV<T>::pack(o));
    if (s2 != nullptr) {       // the second excitation of a two-gate sum sees the same sum of g * x
      EV<T>::unpack(ldv<T>(s2, v), z);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float q = se_sigmoid(z[i]); o[i] = acc[i] * q * (1.0f - q); }
      stv<T>(ds2, v, EV<T>::pack(o));
    }
  }
}
int se_dgate_shares(int HW) {        // pixel shares of the s-side reduction: >= 64 pixels each, at most 64 shares
  int k = HW / 64;
  return k < 1 ? 1 : (k > 64 ? 64 : k);
}
hipError_t launch_se_dgate(int dtype, const void* g, const void* x, const void* s, void* ds, float* part, int N, int HW, int C, hipStream_t st,
                           const void* s2, void* ds2) {
  OCTSEG_NO_F16(dtype);
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0) return hipErrorInvalidValue;
  const int vpc = C / vec, shares = se_dgate_shares(HW);
  if (dtype == DT_F32) hipLaunchKernelGGL(se_dgate_part_kernel<float>, dim3((vpc + SE_CH - 1) / SE_CH, shares, N), dim3(256), 0, st, g, x, part, HW, vpc);
  else hipLaunchKernelGGL(se_dgate_part_kernel<bf16_t>, dim3((vpc + SE_CH - 1) / SE_CH, shares, N), dim3(256), 0, st, g, x, part, HW, vpc);
  const int nvec = N * vpc;
  if (dtype == DT_F32) hipLaunchKernelGGL(se_dgate_fin_kernel<float>, dim3(grid_for((size_t)nvec, 256)), dim3(256), 0, st, part, s, ds, shares, vpc, nvec, s2, ds2);
  else hipLaunchKernelGGL(se_dgate_fin_kernel<bf16_t>, dim3(grid_for((size_t)nvec, 256)), dim3(256), 0, st, part, s, ds, shares, vpc, nvec, s2, ds2);
  return hipGetLastError();
}

}  // namespace octseg
