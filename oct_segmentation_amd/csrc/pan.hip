// pan.hip -- what smp's PAN decoder (decoders/pan; reference sweep configs/tune.yaml:18 ``PAN`` through smp.create_model,
// src/models/smp/model.py:38-44) needs beside the conv kernels: the feature-pyramid-attention block's SINGLE-CHANNEL pyramid.
//
//   x1 = CBR7(maxpool2(x))  x2 = CBR5(maxpool2(x1))  x3 = CBR3(CBR3(maxpool2(x2)))            CBRk = conv k x k (bias) + BatchNorm + ReLU, 1 channel
//   u  = up(up(up(x3) + CBR5(x2)) + CBR7(x1))                                                  up = bilinear, align_corners=True
//   out = u * mid + b1                                                                          mid, b1: 32-channel branches (conv kernels)
//
// Only CBR7(maxpool2(x)) touches the wide feature (2048 channels -> 1): a per-pixel dot product over 49 x C values (fpa_in_*).  Everything
// behind it lives on one-channel maps of at most N x 22 x 22 values: the whole chain -- six BatchNorms over the batch, three conv sizes, two
// max-pools, three resizes -- runs as ONE workgroup in f32 (fpa_pyr_fwd / fpa_pyr_bwd), phase after phase with workgroup barriers, sums in
// double in a fixed order.  One-channel tensors have no place in the 8-channel NHWC vector kernels.
#include "common.h"
#include "ev.h"
#include "kernels.h"

namespace octseg {

template <typename T> static __device__ __forceinline__ float pn_ld(const void* p, size_t i) {
  if (sizeof(T) == 4) return ((const float*)p)[i];
  return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16);
}
template <> __device__ __forceinline__ float pn_ld<f16_t>(const void* p, size_t i) { return (float)__builtin_bit_cast(_Float16, ((const unsigned short*)p)[i]); }
template <typename T> static __device__ __forceinline__ void pn_st(void* p, size_t i, float v) {
  if (sizeof(T) == 4) ((float*)p)[i] = v;
  else ((unsigned short*)p)[i] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
}
template <> __device__ __forceinline__ void pn_st<f16_t>(void* p, size_t i, float v) { const _Float16 h = (_Float16)v; ((unsigned short*)p)[i] = __builtin_bit_cast(unsigned short, h); }

#define PN_DISPATCH(KERNEL, grid, block, ...)                                                          \
  do {                                                                                                 \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, block, 0, st, __VA_ARGS__);      \
    else if (dtype == DT_F16) hipLaunchKernelGGL(KERNEL<f16_t>, grid, block, 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, 0, st, __VA_ARGS__);                     \
  } while (0)

// ------------------------------------------------------------------ MaxPool2d(2, 2) on NHWC vectors, and its gradient (first maximum in scan order, as torch)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const void* x, void* p, const void* dp, void* dx, int H, int W, int vpc, size_t nvec, int accum) {
  constexpr int VEC = EV<T>::VEC;
  const int OH = H / 2, OW = W / 2;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t q = v / vpc;
    const int ox = (int)(q % OW); q /= OW;
    const int oy = (int)(q % OH);
    const size_t n = q / OH;
    float f[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) EV<T>::unpack(ldv<T>(x, ((n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * vpc + cv), f[k]);
    if (dp == nullptr) {
      float m[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { m[i] = f[0][i]; for (int k = 1; k < 4; ++k) if (f[k][i] > m[i] || f[k][i] != f[k][i]) m[i] = f[k][i]; }
      stv<T>(p, v, EV<T>::pack(m));
    } else {
      float g[VEC], o[4][VEC];
      EV<T>::unpack(ldv<T>(dp, v), g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        int am = 0; float m = f[0][i];
        for (int k = 1; k < 4; ++k) if (f[k][i] > m || f[k][i] != f[k][i]) { m = f[k][i]; am = k; }
        for (int k = 0; k < 4; ++k) o[k][i] = k == am ? g[i] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t idx = ((n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * vpc + cv;
        if (accum) { float old[VEC]; EV<T>::unpack(ldv<T>(dx, idx), old);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[k][i] += old[i]; }
        stv<T>(dx, idx, EV<T>::pack(o[k]));
      }
    }
  }
}
hipError_t launch_maxpool2(int dtype, const void* x, void* p, const void* dp, void* dx, int N, int H, int W, int C, int accum, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (C % vec != 0 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * (H / 2) * (W / 2) * (C / vec);
  PN_DISPATCH(maxpool2_kernel, dim3(grid_for(nvec, 256)), dim3(256), x, p, dp, dx, H, W, C / vec, nvec, accum);
  return hipGetLastError();
}

// ------------------------------------------------------------------ the wide end of the pyramid: K x K conv of p [N][H][W][C] to ONE channel
// y[n][oy][ox] = b + sum_{r,s,c} w[r][s][c] p[n][oy + r - K/2][ox + s - K/2][c]: one workgroup per output pixel
template <typename T>
__global__ __launch_bounds__(256) void fpa_in_fwd_kernel(const void* p, const float* w, const float* b, float* y, int H, int W, int C, int K) {
  __shared__ double red[256];
  const size_t o = blockIdx.x;
  const int ox = (int)(o % W), oy = (int)((o / W) % H);
  const size_t n = o / ((size_t)W * H);
  double acc = 0.0;
  for (int t = 0; t < K * K; ++t) {
    const int iy = oy + t / K - K / 2, ix = ox + t % K - K / 2;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
    const size_t base = ((n * H + iy) * W + ix) * C;
    float a = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) a = fmaf(w[(size_t)t * C + c], pn_ld<T>(p, base + c), a);
    acc += (double)a;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) y[o] = (float)red[0] + b[0];
}
// dp[n][y][x][c] = sum_{r,s} w[r][s][c] dy[n][y - r + K/2][x - s + K/2]       (stored: p has this one consumer)
template <typename T>
__global__ __launch_bounds__(256) void fpa_in_bwd_x_kernel(const float* dy, const float* w, void* dp, int H, int W, int C, int K, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    size_t q = e / C;
    const int x = (int)(q % W); q /= W;
    const int y = (int)(q % H);
    const size_t n = q / H;
    float acc = 0.f;
    for (int t = 0; t < K * K; ++t) {
      const int oy = y - t / K + K / 2, ox = x - t % K + K / 2;
      if ((unsigned)oy >= (unsigned)H || (unsigned)ox >= (unsigned)W) continue;
      acc = fmaf(w[(size_t)t * C + c], dy[(n * H + oy) * W + ox], acc);
    }
    pn_st<T>(dp, e, acc);
  }
}
// dw[t][c] += sum_{n,oy,ox} dy[n][oy][ox] p[n][oy + r - K/2][ox + s - K/2][c]: grid (ceil(C / 256), K * K); db += sum dy (block (0, 0))
template <typename T>
__global__ __launch_bounds__(256) void fpa_in_bwd_w_kernel(const void* p, const float* dy, float* dw, float* db, int N, int H, int W, int C, int K) {
  const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  const int dr = t / K - K / 2, ds = t % K - K / 2;
  if (c < C) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n)
      for (int oy = 0; oy < H; ++oy) {
        const int iy = oy + dr;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int ox = 0; ox < W; ++ox) {
          const int ix = ox + ds;
          if ((unsigned)ix >= (unsigned)W) continue;
          acc = fmaf(dy[((size_t)n * H + oy) * W + ox], pn_ld<T>(p, (((size_t)n * H + iy) * W + ix) * C + c), acc);
        }
      }
    dw[(size_t)t * C + c] += acc;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (size_t i = 0; i < (size_t)N * H * W; ++i) s += (double)dy[i];
    db[0] += (float)s;
  }
}
hipError_t launch_fpa_in_fwd(int dtype, const void* p, const float* w, const float* b, float* y, int N, int H, int W, int C, int K, hipStream_t st) {
  PN_DISPATCH(fpa_in_fwd_kernel, dim3((unsigned)((size_t)N * H * W)), dim3(256), p, w, b, y, H, W, C, K);
  return hipGetLastError();
}
hipError_t launch_fpa_in_bwd(int dtype, const void* p, const float* dy, const float* w, void* dp, float* dw, float* db, int N, int H, int W, int C, int K,
                             hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const size_t total = (size_t)N * H * W * C;
  if (dtype == DT_F32) {
    hipLaunchKernelGGL(fpa_in_bwd_x_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, dy, w, dp, H, W, C, K, total);
    hipLaunchKernelGGL(fpa_in_bwd_w_kernel<float>, dim3((C + 255) / 256, K * K), dim3(256), 0, st, p, dy, dw, db, N, H, W, C, K);
  } else {
    hipLaunchKernelGGL(fpa_in_bwd_x_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, dy, w, dp, H, W, C, K, total);
    hipLaunchKernelGGL(fpa_in_bwd_w_kernel<bf16_t>, dim3((C + 255) / 256, K * K), dim3(256), 0, st, p, dy, dw, db, N, H, W, C, K);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ the one-channel pyramid, one workgroup
namespace {
constexpr int PT = 1024;
struct Map { float* v; int H, W; };           // [N][H][W], N from the args

__device__ double pyr_block_sum(double x, double* red) {
  red[threadIdx.x] = x;
  __syncthreads();
  for (int s = PT / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  const double r = red[0];
  __syncthreads();
  return r;
}
// out = conv k x k (pad k/2) of in + bias
__device__ void pyr_conv(const float* in, float* out, const float* w, float bias, int N, int H, int W, int K) {
  for (int e = threadIdx.x; e < N * H * W; e += PT) {
    const int x = e % W, y = (e / W) % H, n = e / (W * H);
    float acc = bias;
    for (int t = 0; t < K * K; ++t) {
      const int iy = y + t / K - K / 2, ix = x + t % K - K / 2;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) acc = fmaf(w[t], in[(n * H + iy) * W + ix], acc);
    }
    out[e] = acc;
  }
  __syncthreads();
}
// din += conv-transpose of dout;  dw[t] += sum dout * in shifted;  db += sum dout
__device__ void pyr_conv_bwd(const float* in, const float* dout, float* din, const float* w, float* dw, float* db, int N, int H, int W, int K, double* red) {
  if (din != nullptr) {
    for (int e = threadIdx.x; e < N * H * W; e += PT) {
      const int x = e % W, y = (e / W) % H, n = e / (W * H);
      float acc = 0.f;
      for (int t = 0; t < K * K; ++t) {
        const int oy = y - t / K + K / 2, ox = x - t % K + K / 2;
        if ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) acc = fmaf(w[t], dout[(n * H + oy) * W + ox], acc);
      }
      din[e] += acc;
    }
  }
  for (int t = 0; t < K * K; ++t) {
    double s = 0.0;
    for (int e = threadIdx.x; e < N * H * W; e += PT) {
      const int x = e % W, y = (e / W) % H, n = e / (W * H);
      const int iy = y + t / K - K / 2, ix = x + t % K - K / 2;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) s += (double)dout[e] * (double)in[(n * H + iy) * W + ix];
    }
    s = pyr_block_sum(s, red);
    if (threadIdx.x == 0) dw[t] += (float)s;
  }
  double s = 0.0;
  for (int e = threadIdx.x; e < N * H * W; e += PT) s += (double)dout[e];
  s = pyr_block_sum(s, red);
  if (threadIdx.x == 0) db[0] += (float)s;
  __syncthreads();
}
// BatchNorm2d(1) + ReLU over n values: train -> batch statistics (biased variance for the normalisation, unbiased into running_var), stats[0..1] =
// mean, rstd kept for the backward; eval -> running statistics
__device__ void pyr_bn_relu(const float* raw, float* out, int n, const float* gamma, const float* beta, float* rmean, float* rvar, float* stats, int train,
                            double* red) {
  float mean, rstd;
  if (train) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += PT) s += (double)raw[e];
    const double mu = pyr_block_sum(s, red) / n;
    double q = 0.0;
    for (int e = threadIdx.x; e < n; e += PT) { const double d = (double)raw[e] - mu; q += d * d; }
    const double var = pyr_block_sum(q, red) / n;
    mean = (float)mu; rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (threadIdx.x == 0) {
      stats[0] = mean; stats[1] = rstd;
      rmean[0] = 0.9f * rmean[0] + 0.1f * mean;
      rvar[0] = 0.9f * rvar[0] + 0.1f * (float)(n > 1 ? var * n / (n - 1) : var);
    }
  } else {
    mean = rmean[0]; rstd = 1.0f / sqrtf(rvar[0] + 1e-5f);
  }
  const float g = gamma[0], b = beta[0];
  for (int e = threadIdx.x; e < n; e += PT) { const float z = (raw[e] - mean) * rstd * g + b; out[e] = z > 0.f ? z : 0.f; }
  __syncthreads();
}
// gradient through ReLU + BatchNorm: dout (wrt the ReLU output; overwritten by d raw), out = the ReLU output, raw = the conv output
__device__ void pyr_bn_relu_bwd(const float* raw, const float* out, float* dout, int n, const float* gamma, const float* stats, float* dgamma, float* dbeta,
                                double* red) {
  const float mean = stats[0], rstd = stats[1];
  double s1 = 0.0, s2 = 0.0;
  for (int e = threadIdx.x; e < n; e += PT) {
    const float g = out[e] > 0.f ? dout[e] : 0.f;
    s1 += (double)g; s2 += (double)g * ((double)(raw[e] - mean) * rstd);
  }
  s1 = pyr_block_sum(s1, red); s2 = pyr_block_sum(s2, red);
  if (threadIdx.x == 0) { dbeta[0] += (float)s1; dgamma[0] += (float)s2; }
  const double c1 = s1 / n, c2 = s2 / n, A = (double)gamma[0] * rstd;
  for (int e = threadIdx.x; e < n; e += PT) {
    const float g = out[e] > 0.f ? dout[e] : 0.f;
    dout[e] = (float)(A * ((double)g - c1 - ((double)(raw[e] - mean) * rstd) * c2));
  }
  __syncthreads();
}
__device__ void pyr_maxpool(const float* in, float* out, int N, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  for (int e = threadIdx.x; e < N * OH * OW; e += PT) {
    const int x = e % OW, y = (e / OW) % OH, n = e / (OW * OH);
    const float* p = in + (n * H + 2 * y) * W + 2 * x;
    float m = p[0];
    if (p[1] > m) m = p[1];
    if (p[W] > m) m = p[W];
    if (p[W + 1] > m) m = p[W + 1];
    out[e] = m;
  }
  __syncthreads();
}
__device__ void pyr_maxpool_bwd(const float* in, const float* dout, float* din, int N, int H, int W) {   // din += (each window has one owner: no race)
  const int OH = H / 2, OW = W / 2;
  for (int e = threadIdx.x; e < N * OH * OW; e += PT) {
    const int x = e % OW, y = (e / OW) % OH, n = e / (OW * OH);
    const int base = (n * H + 2 * y) * W + 2 * x;
    const int offs[4] = {0, 1, W, W + 1};
    int am = 0; float m = in[base];
    for (int k = 1; k < 4; ++k) if (in[base + offs[k]] > m) { m = in[base + offs[k]]; am = k; }
    din[base + offs[am]] += dout[e];
  }
  __syncthreads();
}
// F.interpolate(mode='bilinear', align_corners=True), torch's index rule (scale = (in - 1) / (out - 1) in float, 0 for out == 1)
__device__ __forceinline__ void pyr_bil_coord(int o, int in, int out, int& i0, int& i1, float& l) {
  const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float f = sc * o;
  i0 = (int)f; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l = f - (float)i0;
}
__device__ void pyr_resize(const float* in, float* out, int N, int IH, int IW, int OH, int OW) {
  for (int e = threadIdx.x; e < N * OH * OW; e += PT) {
    const int x = e % OW, y = (e / OW) % OH, n = e / (OW * OH);
    int y0, y1, x0, x1; float ly, lx;
    pyr_bil_coord(y, IH, OH, y0, y1, ly); pyr_bil_coord(x, IW, OW, x0, x1, lx);
    const float* p = in + n * IH * IW;
    out[e] = (1.f - ly) * ((1.f - lx) * p[y0 * IW + x0] + lx * p[y0 * IW + x1]) + ly * ((1.f - lx) * p[y1 * IW + x0] + lx * p[y1 * IW + x1]);
  }
  __syncthreads();
}
__device__ void pyr_resize_bwd(const float* dout, float* din, int N, int IH, int IW, int OH, int OW) {   // din += adjoint, GATHER form: one owner per element,
  // fixed summation order (no atomics: deterministic).  A source pixel is touched only by outputs whose two taps straddle it.
  const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  for (int e = threadIdx.x; e < N * IH * IW; e += PT) {
    const int ix = e % IW, iy = (e / IW) % IH, n = e / (IW * IH);
    int oy0 = 0, oy1 = OH - 1, ox0 = 0, ox1 = OW - 1;
    if (sy > 0.f) { oy0 = max(0, (int)floorf((float)(iy - 1) / sy) - 1); oy1 = min(OH - 1, (int)ceilf((float)(iy + 1) / sy) + 1); }
    if (sx > 0.f) { ox0 = max(0, (int)floorf((float)(ix - 1) / sx) - 1); ox1 = min(OW - 1, (int)ceilf((float)(ix + 1) / sx) + 1); }
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int y0, y1; float ly;
      pyr_bil_coord(oy, IH, OH, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int x0, x1; float lx;
        pyr_bil_coord(ox, IW, OW, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx != 0.f) acc = fmaf(wy * wx, dout[(n * OH + oy) * OW + ox], acc);
      }
    }
    din[e] += acc;
  }
  __syncthreads();
}
}  // namespace

// Scratch layout (floats), level sizes n1 = N h1 w1 (h1 = h / 2), n2 = N h2 w2 (h2 = h1 / 2), n3 = N h3 w3, n0 = N h w:
//   level 1: r1 (x1raw, from fpa_in_fwd), x1, c1 (conv1 raw), y1 (conv1 out), tu     level 2: p2, r2, x2, c2 (conv2 raw), y2, x3u, t
//   level 3: p3, r3a, x3a, r3b, x3                                                      level 0: uu
//   stats: 6 x (mean, rstd) in layer order down1, down2, down3.1, down3.2, conv2, conv1
__global__ __launch_bounds__(1024) void fpa_pyr_fwd_kernel(const FpaPyrArgs a) {
  __shared__ double red[PT];
  const int N = a.N, h1 = a.h / 2, w1 = a.w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  const int n1 = N * h1 * w1, n2 = N * h2 * w2, n3 = N * h3 * w3;
  float* L1 = a.scratch; float* L2 = L1 + 5 * (size_t)n1; float* L3 = L2 + 7 * (size_t)n2; float* UU = L3 + 5 * (size_t)n3; float* ST = UU + (size_t)N * a.h * a.w;
  float *r1 = L1, *x1 = L1 + n1, *c1 = L1 + 2 * n1, *y1 = L1 + 3 * n1, *tu = L1 + 4 * n1;
  float *p2 = L2, *r2 = L2 + n2, *x2 = L2 + 2 * n2, *c2 = L2 + 3 * n2, *y2 = L2 + 4 * n2, *x3u = L2 + 5 * n2, *t = L2 + 6 * n2;
  float *p3 = L3, *r3a = L3 + n3, *x3a = L3 + 2 * n3, *r3b = L3 + 3 * n3, *x3 = L3 + 4 * n3;
  // layer l: conv weight a.w_[l] ([k*k]), bias a.b_[l], BatchNorm gamma / beta a.g_[l] / a.be_[l], running a.rm_[l] / a.rv_[l]
  pyr_bn_relu(r1, x1, n1, a.g_[0], a.be_[0], a.rm_[0], a.rv_[0], ST + 0, a.train, red);                      // down1's BatchNorm + ReLU (its conv: fpa_in_fwd)
  pyr_maxpool(x1, p2, N, h1, w1);
  pyr_conv(p2, r2, a.w_[1], a.b_[1][0], N, h2, w2, 5);
  pyr_bn_relu(r2, x2, n2, a.g_[1], a.be_[1], a.rm_[1], a.rv_[1], ST + 2, a.train, red);
  pyr_maxpool(x2, p3, N, h2, w2);
  pyr_conv(p3, r3a, a.w_[2], a.b_[2][0], N, h3, w3, 3);
  pyr_bn_relu(r3a, x3a, n3, a.g_[2], a.be_[2], a.rm_[2], a.rv_[2], ST + 4, a.train, red);
  pyr_conv(x3a, r3b, a.w_[3], a.b_[3][0], N, h3, w3, 3);
  pyr_bn_relu(r3b, x3, n3, a.g_[3], a.be_[3], a.rm_[3], a.rv_[3], ST + 6, a.train, red);
  pyr_resize(x3, x3u, N, h3, w3, h2, w2);
  pyr_conv(x2, c2, a.w_[4], a.b_[4][0], N, h2, w2, 5);
  pyr_bn_relu(c2, y2, n2, a.g_[4], a.be_[4], a.rm_[4], a.rv_[4], ST + 8, a.train, red);
  for (int e = threadIdx.x; e < n2; e += PT) t[e] = y2[e] + x3u[e];
  __syncthreads();
  pyr_resize(t, tu, N, h2, w2, h1, w1);
  pyr_conv(x1, c1, a.w_[5], a.b_[5][0], N, h1, w1, 7);
  pyr_bn_relu(c1, y1, n1, a.g_[5], a.be_[5], a.rm_[5], a.rv_[5], ST + 10, a.train, red);
  for (int e = threadIdx.x; e < n1; e += PT) tu[e] += y1[e];                                                 // u = up(t) + conv1(x1)
  __syncthreads();
  pyr_resize(tu, UU, N, h1, w1, a.h, a.w);
}
// Gradient scratch `g` (floats): d1 [n1] x 3 (du / dy1, dx1, dr1), d2 [n2] x 4, d3 [n3] x 3, zeroed by the launcher.  Input: a.duu [N][h][w].
// Output: dr1 (gradient wrt x1raw, for fpa_in_bwd) in g[2 n1 ..], parameter gradients accumulated into a.dw_ / db_ / dg_ / dbe_.
__global__ __launch_bounds__(1024) void fpa_pyr_bwd_kernel(const FpaPyrArgs a) {
  __shared__ double red[PT];
  const int N = a.N, h1 = a.h / 2, w1 = a.w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  const int n1 = N * h1 * w1, n2 = N * h2 * w2, n3 = N * h3 * w3;
  float* L1 = a.scratch; float* L2 = L1 + 5 * (size_t)n1; float* L3 = L2 + 7 * (size_t)n2; float* UU = L3 + 5 * (size_t)n3; float* ST = UU + (size_t)N * a.h * a.w;
  float *r1 = L1, *x1 = L1 + n1, *c1 = L1 + 2 * n1, *y1 = L1 + 3 * n1;
  float *p2 = L2, *r2 = L2 + n2, *x2 = L2 + 2 * n2, *c2 = L2 + 3 * n2, *y2 = L2 + 4 * n2;
  float *p3 = L3, *r3a = L3 + n3, *x3a = L3 + 2 * n3, *r3b = L3 + 3 * n3, *x3 = L3 + 4 * n3;
  float* G = a.gscratch;
  float *du = G, *dx1 = G + n1, *dy1 = G + 2 * n1;
  float *dt = G + 3 * n1, *dx2 = dt + n2, *dy2 = dt + 2 * n2, *dp2 = dt + 3 * n2;
  float *dx3 = dt + 4 * n2, *dx3a = dx3 + n3, *dp3 = dx3 + 2 * n3;
  // uu = up(u)
  pyr_resize_bwd(a.duu, du, N, h1, w1, a.h, a.w);
  // u = up(t) + y1, y1 = relu(bn(c1)), c1 = conv7(x1)
  for (int e = threadIdx.x; e < n1; e += PT) dy1[e] = du[e];
  __syncthreads();
  pyr_bn_relu_bwd(c1, y1, dy1, n1, a.g_[5], ST + 10, a.dg_[5], a.dbe_[5], red);
  pyr_conv_bwd(x1, dy1, dx1, a.w_[5], a.dw_[5], a.db_[5], N, h1, w1, 7, red);
  pyr_resize_bwd(du, dt, N, h2, w2, h1, w1);
  // t = y2 + x3u, y2 = relu(bn(c2)), c2 = conv5(x2); x3u = up(x3)
  for (int e = threadIdx.x; e < n2; e += PT) dy2[e] = dt[e];
  __syncthreads();
  pyr_bn_relu_bwd(c2, y2, dy2, n2, a.g_[4], ST + 8, a.dg_[4], a.dbe_[4], red);
  pyr_conv_bwd(x2, dy2, dx2, a.w_[4], a.dw_[4], a.db_[4], N, h2, w2, 5, red);
  pyr_resize_bwd(dt, dx3, N, h3, w3, h2, w2);
  // x3 = relu(bn(r3b)), r3b = conv3(x3a); x3a = relu(bn(r3a)), r3a = conv3(p3); p3 = maxpool(x2)
  pyr_bn_relu_bwd(r3b, x3, dx3, n3, a.g_[3], ST + 6, a.dg_[3], a.dbe_[3], red);
  pyr_conv_bwd(x3a, dx3, dx3a, a.w_[3], a.dw_[3], a.db_[3], N, h3, w3, 3, red);
  pyr_bn_relu_bwd(r3a, x3a, dx3a, n3, a.g_[2], ST + 4, a.dg_[2], a.dbe_[2], red);
  pyr_conv_bwd(p3, dx3a, dp3, a.w_[2], a.dw_[2], a.db_[2], N, h3, w3, 3, red);
  pyr_maxpool_bwd(x2, dp3, dx2, N, h2, w2);
  // x2 = relu(bn(r2)), r2 = conv5(p2), p2 = maxpool(x1)
  pyr_bn_relu_bwd(r2, x2, dx2, n2, a.g_[1], ST + 2, a.dg_[1], a.dbe_[1], red);
  pyr_conv_bwd(p2, dx2, dp2, a.w_[1], a.dw_[1], a.db_[1], N, h2, w2, 5, red);
  pyr_maxpool_bwd(x1, dp2, dx1, N, h1, w1);
  // x1 = relu(bn(r1)): dx1 -> d r1 in place (r1 = the wide 7x7 conv's output)
  pyr_bn_relu_bwd(r1, x1, dx1, n1, a.g_[0], ST + 0, a.dg_[0], a.dbe_[0], red);
}
size_t fpa_pyr_scratch_floats(int N, int h, int w) {
  const int h1 = h / 2, w1 = w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  return (size_t)5 * N * h1 * w1 + (size_t)7 * N * h2 * w2 + (size_t)5 * N * h3 * w3 + (size_t)N * h * w + 16;
}
size_t fpa_pyr_gscratch_floats(int N, int h, int w) {
  const int h1 = h / 2, w1 = w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  return (size_t)3 * N * h1 * w1 + (size_t)4 * N * h2 * w2 + (size_t)3 * N * h3 * w3;
}
hipError_t launch_fpa_pyr_fwd(const FpaPyrArgs& a, hipStream_t st) {
  if (a.h / 8 < 1 || a.w / 8 < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fpa_pyr_fwd_kernel, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_fpa_pyr_bwd(const FpaPyrArgs& a, hipStream_t st) {
  hipError_t e = hipMemsetAsync(a.gscratch, 0, fpa_pyr_gscratch_floats(a.N, a.h, a.w) * sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fpa_pyr_bwd_kernel, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ out = uu * mid + b1, and its gradients
// forward: out[n][p][c] = uu[n][p] mid[n][p][c] + b1[n][c];  backward: dmid = g uu, duu[n][p] = sum_c g mid   (d b1 = per-image sums of g: image_sum)
template <typename T>
__global__ __launch_bounds__(256) void fpa_mix_kernel(const float* uu, const void* mid, const void* b1, void* out, const void* g, void* dmid, float* duu, int HW,
                                                      int C, size_t npix) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW;
    const float u = uu[p];
    if (g == nullptr) {
      for (int c = 0; c < C; ++c) pn_st<T>(out, p * C + c, fmaf(u, pn_ld<T>(mid, p * C + c), pn_ld<T>(b1, n * C + c)));
    } else {
      float s = 0.f;
      for (int c = 0; c < C; ++c) {
        const float gv = pn_ld<T>(g, p * C + c);
        s = fmaf(gv, pn_ld<T>(mid, p * C + c), s);
        pn_st<T>(dmid, p * C + c, gv * u);
      }
      duu[p] = s;
    }
  }
}
hipError_t launch_fpa_mix(int dtype, const float* uu, const void* mid, const void* b1, void* out, const void* g, void* dmid, float* duu, int N, int HW, int C,
                          hipStream_t st) {
  const size_t npix = (size_t)N * HW;
  PN_DISPATCH(fpa_mix_kernel, dim3(grid_for(npix, 256)), dim3(256), uu, mid, b1, out, g, dmid, duu, HW, C, npix);
  return hipGetLastError();
}

// out = a + b (GAU: y_up + z)
template <typename T>
__global__ __launch_bounds__(256) void add2_kernel(const void* a, const void* b, void* out, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float x[VEC], y[VEC];
    EV<T>::unpack(ldv<T>(a, v), x);
    EV<T>::unpack(ldv<T>(b, v), y);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += y[i];
    stv<T>(out, v, EV<T>::pack(x));
  }
}
hipError_t launch_add2(int dtype, const void* a, const void* b, void* out, size_t numel, hipStream_t st) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  if (numel % vec != 0) return hipErrorInvalidValue;
  PN_DISPATCH(add2_kernel, dim3(grid_for(numel / vec, 256)), dim3(256), a, b, out, numel / vec);
  return hipGetLastError();
}

}  // namespace octseg
