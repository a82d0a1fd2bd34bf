// effnet.hip -- what the EfficientNet encoders (efficientnet_pytorch's MBConvBlock as smp 0.3.3's EfficientNetEncoder runs it; reference sweep
// configs/tune.yaml:25-28 -> smp.create_model(arch, 'efficientnet-b0' | '-b5' | '-b7'), src/models/smp/model.py:38-44) need beside the conv
// kernels.  All of it is HBM-bound NHWC sweep work on 16-byte channel vectors, f32 arithmetic:
//   dwg_*      depthwise k x k (k = 3, 5) stride 1 / 2 convolution with TF-"same" STATIC padding (top/left pad given; the bottom/right pad is
//              whatever the output size implies): forward, data gradient (gather form, no atomics), weight gradient (per-tap sums; partials
//              per workgroup, combined with float atomics -- one workgroup per channel chunk in deterministic mode)
//   bnx_*      out = act(y * scale + shift) * dscale[n] + post with act = identity | swish (x sigmoid(x)), dscale = the drop_connect factor of
//              the block's id-skip (nullable), and the matching gradient wrt the BatchNorm OUTPUT (the BatchNorm backward itself is the
//              existing bn_bwd_* sweep with mask 0)
//   sefc_*     the squeeze-excite excitation s = W2 swish(W1 m + b1) + b2 on pooled vectors m [N][C]: reduction widths of 4 .. 160 channels do
//              not fit the 8-channel vectors of the conv kernels, and the whole thing is 10^5 MACs per image -- one workgroup per image
#include "common.h"
#include "ev.h"
#include "kernels.h"

#include <algorithm>

namespace octseg {

static __device__ __forceinline__ float ef_sigmoid(float z) {
  const float e = expf(-fabsf(z));
  return z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}
template <typename T> static __device__ __forceinline__ float ef_ld1(const void* p, size_t i);
template <> __device__ __forceinline__ float ef_ld1<float>(const void* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float ef_ld1<bf16_t>(const void* p, size_t i) { return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16); }
template <> __device__ __forceinline__ float ef_ld1<f16_t>(const void* p, size_t i) { return (float)__builtin_bit_cast(_Float16, ((const unsigned short*)p)[i]); }
template <typename T> static __device__ __forceinline__ void ef_st1(void* p, size_t i, float v);
template <> __device__ __forceinline__ void ef_st1<float>(void* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void ef_st1<bf16_t>(void* p, size_t i, float v) { ((unsigned short*)p)[i] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ void ef_st1<f16_t>(void* p, size_t i, float v) { const _Float16 h = (_Float16)v; ((unsigned short*)p)[i] = __builtin_bit_cast(unsigned short, h); }

#define EF_DISPATCH(KERNEL, grid, ...)                                                          \
  do {                                                                                          \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, st, __VA_ARGS__);      \
    else if (dtype == DT_F16) hipLaunchKernelGGL(KERNEL<f16_t>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)
#define EF_DISPATCH_TRAIN(KERNEL, grid, ...)                                                    \
  do {                                                                                          \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, st, __VA_ARGS__);      \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)

// ------------------------------------------------------------------ depthwise k x k, stride s, static "same" padding
// out[n][oy][ox][c] = sum_{r,s} w[r][s][c] * in[n][oy * st - pt + r][ox * st - pt + s][c]     (zero outside the input)
template <typename T>
__global__ __launch_bounds__(256) void dwg_fwd_kernel(const DwgArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = (size_t)a.N * a.OH * a.OW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % a.OW); p /= a.OW;
    const int oy = (int)(p % a.OH);
    const size_t n = p / a.OH;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int r = 0; r < a.K; ++r) {
      const int iy = oy * a.stride - a.pad + r;
      if ((unsigned)iy >= (unsigned)a.H) continue;
      for (int s = 0; s < a.K; ++s) {
        const int ix = ox * a.stride - a.pad + s;
        if ((unsigned)ix >= (unsigned)a.W) continue;
        float f[VEC];
        EV<T>::unpack(ldv<T>(a.in, ((n * a.H + iy) * a.W + ix) * vpc + cv), f);
        const float* w = a.w + (size_t)(r * a.K + s) * a.C + cv * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[i], f[i], acc[i]);
      }
    }
    stv<T>(a.out, v, EV<T>::pack(acc));
  }
}
// gin[n][iy][ix][c] (+)= sum over (r, s) with (iy + pt - r) = st * oy, (ix + pt - s) = st * ox of w[r][s][c] * gout[n][oy][ox][c]
template <typename T>
__global__ __launch_bounds__(256) void dwg_bwd_data_kernel(const DwgArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = (size_t)a.N * a.H * a.W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ix = (int)(p % a.W); p /= a.W;
    const int iy = (int)(p % a.H);
    const size_t n = p / a.H;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int r = 0; r < a.K; ++r) {
      const int ty = iy + a.pad - r;
      if (ty < 0 || ty % a.stride != 0) continue;
      const int oy = ty / a.stride;
      if (oy >= a.OH) continue;
      for (int s = 0; s < a.K; ++s) {
        const int tx = ix + a.pad - s;
        if (tx < 0 || tx % a.stride != 0) continue;
        const int ox = tx / a.stride;
        if (ox >= a.OW) continue;
        float f[VEC];
        EV<T>::unpack(ldv<T>(a.out, ((n * a.OH + oy) * a.OW + ox) * vpc + cv), f);     // (a.out = gout here)
        const float* w = a.w + (size_t)(r * a.K + s) * a.C + cv * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[i], f[i], acc[i]);
      }
    }
    if (a.accum) {
      float o[VEC];
      EV<T>::unpack(ldv<T>(a.gin, v), o);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += o[i];
    }
    stv<T>(a.gin, v, EV<T>::pack(acc));
  }
}
// dw[r][s][c] += sum_{n, oy, ox} gout[n][oy][ox][c] * in[n][oy * st - pt + r][ox * st - pt + s][c]: one kernel row r per launch-z,
// K tap accumulators per thread; block (x: channel chunk of 32 vectors, y: pixel share)
constexpr int DWG_CH = 32;
template <typename T>
__global__ __launch_bounds__(256) void dwg_bwd_w_kernel(const DwgArgs a) {
  constexpr int VEC = EV<T>::VEC;
  constexpr int KMAX = 5;
  __shared__ float red[256 * VEC];
  const int vpc = a.C / VEC;
  const int v0 = blockIdx.x * DWG_CH, nv = min(DWG_CH, vpc - v0), rows = 256 / nv;
  const int rl = threadIdx.x / nv, cv = threadIdx.x - rl * nv;
  const int r = blockIdx.z;
  float acc[KMAX][VEC];
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[s][i] = 0.f;
  const size_t npix = (size_t)a.N * a.OH * a.OW;
  if (rl < rows)
    for (size_t p = (size_t)blockIdx.y * rows + rl; p < npix; p += (size_t)gridDim.y * rows) {
      size_t q = p;
      const int ox = (int)(q % a.OW); q /= a.OW;
      const int oy = (int)(q % a.OH);
      const size_t n = q / a.OH;
      const int iy = oy * a.stride - a.pad + r;
      if ((unsigned)iy >= (unsigned)a.H) continue;
      float g[VEC];
      EV<T>::unpack(ldv<T>(a.out, p * vpc + v0 + cv), g);                       // (a.out = gout)
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        if (s < a.K) {
          const int ix = ox * a.stride - a.pad + s;
          if ((unsigned)ix < (unsigned)a.W) {
            float f[VEC];
            EV<T>::unpack(ldv<T>(a.in, ((n * a.H + iy) * a.W + ix) * vpc + v0 + cv), f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[s][i] = fmaf(g[i], f[i], acc[s][i]);
          }
        }
      }
    }
  for (int s = 0; s < a.K; ++s) {
    __syncthreads();
    if (rl < rows)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[(rl * nv + cv) * VEC + i] = acc[s][i];
    __syncthreads();
    if (rl == 0) {
      float t[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) t[i] = red[cv * VEC + i];
      for (int k = 1; k < rows; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) t[i] += red[(k * nv + cv) * VEC + i];
      float* d = a.dw + (size_t)(r * a.K + s) * a.C + (v0 + cv) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) atomicAdd(d + i, t[i]);
    }
  }
}
static bool dwg_ok(int dtype, const DwgArgs& a) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  return a.C % vec == 0 && (a.K == 3 || a.K == 5) && (a.stride == 1 || a.stride == 2) && a.pad >= 0 && a.pad < a.K && a.OH >= 1 && a.OW >= 1 &&
         (a.OH - 1) * a.stride - a.pad < a.H && (a.OW - 1) * a.stride - a.pad < a.W;
}
hipError_t launch_dwg_fwd(int dtype, const DwgArgs& a, hipStream_t st) {
  if (!dwg_ok(dtype, a)) return hipErrorInvalidValue;
  const size_t nvec = (size_t)a.N * a.OH * a.OW * (a.C / (dtype == DT_F32 ? 4 : 8));
  EF_DISPATCH(dwg_fwd_kernel, dim3(grid_for(nvec, 256)), a);
  return hipGetLastError();
}
hipError_t launch_dwg_bwd_data(int dtype, const DwgArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  if (!dwg_ok(dtype, a)) return hipErrorInvalidValue;
  const size_t nvec = (size_t)a.N * a.H * a.W * (a.C / (dtype == DT_F32 ? 4 : 8));
  EF_DISPATCH_TRAIN(dwg_bwd_data_kernel, dim3(grid_for(nvec, 256)), a);
  return hipGetLastError();
}
hipError_t launch_dwg_bwd_w(int dtype, const DwgArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  if (!dwg_ok(dtype, a)) return hipErrorInvalidValue;
  const int vpc = a.C / (dtype == DT_F32 ? 4 : 8);
  const size_t npix = (size_t)a.N * a.OH * a.OW;
  int shares = deterministic_mode() ? 1 : (int)std::min<size_t>(256, (npix + 255) / 256);   // one writer per weight in deterministic mode
  if (shares < 1) shares = 1;
  EF_DISPATCH_TRAIN(dwg_bwd_w_kernel, dim3((vpc + DWG_CH - 1) / DWG_CH, shares, a.K), a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ act(bn(y)) * dscale[n] + post, and its gradient wrt bn(y)
// act: 0 identity, 1 swish.  scale == nullptr: the BatchNorm is already applied (folded into the producing conv's epilogue, eval).
template <typename T>
__global__ __launch_bounds__(256) void bnx_fwd_kernel(const BnxArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = a.npix * (size_t)vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % vpc) * VEC;
    float x[VEC];
    EV<T>::unpack(ldv<T>(a.y, v), x);
    if (a.scale) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = fmaf(x[i], a.scale[c + i], a.shift[c + i]);
    }
    if (a.act == 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = x[i] * ef_sigmoid(x[i]);
    }
    if (a.dscale) {
      const float d = a.dscale[v / ((size_t)a.hw * vpc)];
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] *= d;
    }
    if (a.post) {
      float pp[VEC];
      EV<T>::unpack(ldv<T>(a.post, v), pp);
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] += pp[i];
    }
    stv<T>(a.out, v, EV<T>::pack(x));
  }
}
// gz = g * dscale[n] * act'(y * scale + shift)      (out = gz; post's own gradient is g itself: the caller accumulates it)
template <typename T>
__global__ __launch_bounds__(256) void bnx_bwd_kernel(const BnxArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = a.npix * (size_t)vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % vpc) * VEC;
    float g[VEC];
    EV<T>::unpack(ldv<T>(a.post, v), g);          // (a.post = g here)
    if (a.dscale) {
      const float d = a.dscale[v / ((size_t)a.hw * vpc)];
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] *= d;
    }
    if (a.act == 1) {
      float y[VEC];
      EV<T>::unpack(ldv<T>(a.y, v), y);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float z = fmaf(y[i], a.scale[c + i], a.shift[c + i]);
        const float sg = ef_sigmoid(z);
        g[i] *= sg * (1.0f + z * (1.0f - sg));     // d/dz z sigmoid(z)
      }
    }
    stv<T>(a.out, v, EV<T>::pack(g));
  }
}
static inline int ef_grid_for_channels(size_t nvec, int vpc) {   // the stride (grid * 256) must be a multiple of vpc: c is loop invariant only then
  int g = grid_for(nvec, 256);
  int x = vpc, y = 256;
  while (y) { const int t = x % y; x = y; y = t; }
  const int m = vpc / x;
  g = (g / m) * m;
  return g < m ? m : g;
}
hipError_t launch_bnx_fwd(int dtype, const BnxArgs& a, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (a.C % vec != 0) return hipErrorInvalidValue;
  EF_DISPATCH(bnx_fwd_kernel, dim3(grid_for(a.npix * (size_t)(a.C / vec), 256)), a);
  return hipGetLastError();
}
hipError_t launch_bnx_bwd(int dtype, const BnxArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (a.C % vec != 0) return hipErrorInvalidValue;
  EF_DISPATCH_TRAIN(bnx_bwd_kernel, dim3(grid_for(a.npix * (size_t)(a.C / vec), 256)), a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ squeeze-excite excitation on pooled vectors
// s[n][c] = b2[c] + sum_j W2[c][j] swish(h[n][j]),  h[n][j] = b1[j] + sum_c W1[j][c] m[n][c].   One workgroup per image; h kept (float).
template <typename T>
__global__ __launch_bounds__(256) void sefc_fwd_kernel(const SefcArgs a) {
  extern __shared__ float sm[];            // m [C], act [R]
  float* m = sm; float* act = sm + a.C;
  const size_t n = blockIdx.x;
  for (int c = threadIdx.x; c < a.C; c += 256) m[c] = ef_ld1<T>(a.m, n * a.C + c);
  __syncthreads();
  for (int j = threadIdx.x; j < a.R; j += 256) {
    float h = a.b1[j];
    const float* w = a.w1 + (size_t)j * a.C;
    for (int c = 0; c < a.C; ++c) h = fmaf(w[c], m[c], h);
    a.h[n * a.R + j] = h;
    act[j] = a.act ? h * ef_sigmoid(h) : fmaxf(h, 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float s = a.b2[c];
    const float* w = a.w2 + (size_t)c * a.R;
    for (int j = 0; j < a.R; ++j) s = fmaf(w[j], act[j], s);
    ef_st1<T>(a.s, n * a.C + c, s);
  }
}
// per image: dh[n][j] = swish'(h) * sum_c W2[c][j] ds[n][c] (kept, float);  dm[n][c] = sum_j W1[j][c] dh[n][j]
template <typename T>
__global__ __launch_bounds__(256) void sefc_bwd_kernel(const SefcArgs a) {
  extern __shared__ float sm[];            // ds [C], dh [R]
  float* ds = sm; float* dh = sm + a.C;
  const size_t n = blockIdx.x;
  for (int c = threadIdx.x; c < a.C; c += 256) ds[c] = ef_ld1<T>(a.ds, n * a.C + c);
  __syncthreads();
  for (int j = threadIdx.x; j < a.R; j += 256) {
    float d = 0.f;
    for (int c = 0; c < a.C; ++c) d = fmaf(a.w2[(size_t)c * a.R + j], ds[c], d);
    const float h = a.h[n * a.R + j], sg = ef_sigmoid(h);
    d *= a.act ? sg * (1.0f + h * (1.0f - sg)) : (h > 0.f ? 1.f : 0.f);
    dh[j] = d;
    a.dh[n * a.R + j] = d;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float d = 0.f;
    for (int j = 0; j < a.R; ++j) d = fmaf(a.w1[(size_t)j * a.C + c], dh[j], d);
    ef_st1<T>(a.dm, n * a.C + c, d);
  }
}
// weight gradients, one thread per element, images summed in order (deterministic):
// dW2[c][j] += sum_n ds[n][c] swish(h[n][j]);  db2[c] += sum_n ds[n][c];  dW1[j][c] += sum_n dh[n][j] m[n][c];  db1[j] += sum_n dh[n][j]
template <typename T>
__global__ __launch_bounds__(256) void sefc_wgrad_kernel(const SefcArgs a) {
  const int CR = a.C * a.R;
  const int total = 2 * CR + a.C + a.R;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    float acc = 0.f;
    if (e < CR) {                       // dW2[c][j]
      const int c = e / a.R, j = e - c * a.R;
      for (int n = 0; n < a.N; ++n) {
        const float h = a.h[(size_t)n * a.R + j];
        acc = fmaf(ef_ld1<T>(a.ds, (size_t)n * a.C + c), a.act ? h * ef_sigmoid(h) : fmaxf(h, 0.f), acc);
      }
      a.dw2[e] += acc;
    } else if (e < 2 * CR) {            // dW1[j][c]
      const int k = e - CR, j = k / a.C, c = k - j * a.C;
      for (int n = 0; n < a.N; ++n) acc = fmaf(a.dh[(size_t)n * a.R + j], ef_ld1<T>(a.m, (size_t)n * a.C + c), acc);
      a.dw1[k] += acc;
    } else if (e < 2 * CR + a.C) {      // db2[c]
      const int c = e - 2 * CR;
      for (int n = 0; n < a.N; ++n) acc += ef_ld1<T>(a.ds, (size_t)n * a.C + c);
      a.db2[c] += acc;
    } else {                            // db1[j]
      const int j = e - 2 * CR - a.C;
      for (int n = 0; n < a.N; ++n) acc += a.dh[(size_t)n * a.R + j];
      a.db1[j] += acc;
    }
  }
}
hipError_t launch_sefc_fwd(int dtype, const SefcArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(a.C + a.R) * sizeof(float);
  if (lds > 60 * 1024 || a.R < 1) return hipErrorInvalidValue;
  if (dtype == DT_F32) hipLaunchKernelGGL(sefc_fwd_kernel<float>, dim3(a.N), dim3(256), lds, st, a);
  else if (dtype == DT_F16) hipLaunchKernelGGL(sefc_fwd_kernel<f16_t>, dim3(a.N), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(sefc_fwd_kernel<bf16_t>, dim3(a.N), dim3(256), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_sefc_bwd(int dtype, const SefcArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const size_t lds = (size_t)(a.C + a.R) * sizeof(float);
  if (lds > 60 * 1024 || a.R < 1) return hipErrorInvalidValue;
  if (dtype == DT_F32) hipLaunchKernelGGL(sefc_bwd_kernel<float>, dim3(a.N), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(sefc_bwd_kernel<bf16_t>, dim3(a.N), dim3(256), lds, st, a);
  const int total = 2 * a.C * a.R + a.C + a.R;
  if (dtype == DT_F32) hipLaunchKernelGGL(sefc_wgrad_kernel<float>, dim3(grid_for((size_t)total, 256)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(sefc_wgrad_kernel<bf16_t>, dim3(grid_for((size_t)total, 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace octseg
