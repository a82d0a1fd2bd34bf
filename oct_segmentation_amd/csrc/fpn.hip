// fpn.hip -- the HBM-bound kernels that the FPN decoder adds to the hot path (gfx950): GroupNorm(32) + ReLU with the bilinear x2
// (align_corners=True) resample fused into its apply pass, the adjoint of that resample, GroupNorm backward, the nearest-x2 fill of an
// FPNBlock, merge-add + Dropout2d with an injected keep mask, and the x4 bilinear resample of the head's logits.
//
// Reference: `FPN` is one of the architectures the reference sweeps (configs/tune.yaml:9-18 -> smp.create_model(arch='FPN', ...),
// src/models/smp/model.py:38-44; several per-class winners of eval/tuning/configs_best.xlsx); the arithmetic is smp 0.3.3
// decoders/fpn/decoder.py + torch's upsample_bilinear2d / group_norm, restated in oracle/nets.py (FPNDecoder).
// All tensors NHWC, 16-byte vectors, f32 arithmetic; per-(image, channel) partial sums go through deterministic slabs (no float atomics).
#include "common.h"
#include "ev.h"
#include "kernels.h"

namespace octseg {

// torch's align_corners=True source index: scale = (in - 1) / (out - 1) in float, x = scale * o, i0 = (int)x, lambda1 = x - i0
struct Lerp { int i0, i1; float w0, w1; };
static __device__ __forceinline__ Lerp lerp_of(int o, int in, float scale) {
  const float x = scale * (float)o;
  Lerp l;
  l.i0 = min((int)x, in - 1);
  l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
  l.w1 = x - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}
static inline float lerp_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// ------------------------------------------------------------------ nearest x2 fill (FPNBlock: x = interpolate(x, 2, 'nearest'); the skip conv then accumulates)
template <typename T>
__global__ __launch_bounds__(256) void up2_fill_kernel(const void* in, void* out, int N, int H, int W, int vpc) {
  const size_t nvec = (size_t)N * (2 * H) * (2 * W) * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % (2 * W)); p /= (2 * W);
    const int oy = (int)(p % (2 * H));
    const int n = (int)(p / (2 * H));
    stv<T>(out, v, ldv<T>(in, (((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * vpc + cv));
  }
}
hipError_t launch_up2_fill(int dtype, const void* in, void* out, int N, int H, int W, int C, hipStream_t st) {
  const int vpc = C / (dtype == DT_F32 ? 4 : 8);
  const size_t nvec = (size_t)N * 4 * H * W * vpc;
  const int g = grid_for(nvec, 256);
  if (dtype == DT_F32) hipLaunchKernelGGL(up2_fill_kernel<float>, dim3(g), dim3(256), 0, st, in, out, N, H, W, vpc);
  else hipLaunchKernelGGL(up2_fill_kernel<bf16_t>, dim3(g), dim3(256), 0, st, in, out, N, H, W, vpc);   // (16-byte moves: any 2-byte type)
  return hipGetLastError();
}

// ------------------------------------------------------------------ GroupNorm: per-(image, slab, channel) partial sums
// mode 0 (forward): (sum y, sum y^2).  mode 1 (backward): dz = g * [y * scale + shift > 0], (sum dz, sum dz * xhat), xhat = (y - mean_g) * rstd_g.
// grid (S, N); a thread keeps one channel vector, threads sharing it split the slab's pixels; combined through LDS.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gn_reduce_kernel(const GnArgs a) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256][2 * VEC + 1];
  const int vpc = a.C / VEC;                  // <= 256 (C <= 1024 f32 / 2048 two-byte), a power of two times ...: host-checked to divide 256
  const int tpv = 256 / vpc;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc;
  const int n = blockIdx.y, s = blockIdx.x, S = gridDim.x;
  const int c = cv * VEC;
  const size_t per = (a.HW + S - 1) / S;
  const size_t p0 = (size_t)s * per, p1 = min(a.HW, p0 + per);
  float sc[VEC], sh[VEC], mu[VEC], rs[VEC];
  if (MODE == 1) {
    const float* ss = a.ss + ((size_t)n * a.C + c) * 2;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sc[i] = ss[2 * i]; sh[i] = ss[2 * i + 1];
      const int g = (c + i) / a.cpg;
      mu[i] = a.stat[((size_t)n * a.G + g) * 2]; rs[i] = a.stat[((size_t)n * a.G + g) * 2 + 1];
    }
  }
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  const size_t base = (size_t)n * a.HW;
  for (size_t p = p0 + pl; p < p1; p += tpv) {
    float y[VEC];
    EV<T>::unpack(ldv<T>(a.y, (base + p) * vpc + cv), y);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s1[i] += y[i]; s2[i] += y[i] * y[i]; }
    } else {
      float g[VEC];
      EV<T>::unpack(ldv<T>(a.g, (base + p) * vpc + cv), g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float dz = fmaf(y[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
        s1[i] += dz; s2[i] += dz * (y[i] - mu[i]) * rs[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[threadIdx.x][i] = s1[i]; red[threadIdx.x][VEC + i] = s2[i]; }
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < tpv; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s1[i] += red[threadIdx.x + k * vpc][i]; s2[i] += red[threadIdx.x + k * vpc][VEC + i]; }
    float* o = a.part + (((size_t)n * S + s) * a.C + c) * 2;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { o[2 * i] = s1[i]; o[2 * i + 1] = s2[i]; }
  }
}

// forward finalize: one block per image, one thread per channel: group mean / rstd (biased variance, eps), per-(image, channel)
// scale = gamma * rstd, shift = beta - mean * scale
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const GnArgs a, int S) {
  __shared__ double g1[1024], g2[1024];
  const int n = blockIdx.x, c = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (c < a.C)
    for (int s = 0; s < S; ++s) { const float* p = a.part + (((size_t)n * S + s) * a.C + c) * 2; s1 += (double)p[0]; s2 += (double)p[1]; }
  g1[c] = s1; g2[c] = s2;
  __syncthreads();
  if (c < a.C) {
    const int g = c / a.cpg;
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < a.cpg; ++k) { t1 += g1[g * a.cpg + k]; t2 += g2[g * a.cpg + k]; }
    const double cnt = (double)a.HW * a.cpg;
    const double mean = t1 / cnt;
    double var = t2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float scv = a.gamma[c] * rstd;
    a.ss[((size_t)n * a.C + c) * 2] = scv;
    a.ss[((size_t)n * a.C + c) * 2 + 1] = a.beta[c] - (float)mean * scv;
    if (c % a.cpg == 0) { a.stat[((size_t)n * a.G + g) * 2] = (float)mean; a.stat[((size_t)n * a.G + g) * 2 + 1] = rstd; }
  }
}

// backward finalize: ONE block (threads = channels) walks the images in order: group means of (dz * gamma) and (dz * gamma * xhat) per
// image -> coef[n][g] = (m1, m2); dgamma / dbeta accumulate over the images in index order (deterministic)
__global__ __launch_bounds__(1024) void gn_bwd_finalize_kernel(const GnArgs a, int S, int N) {
  __shared__ double g1[1024], g2[1024];
  const int c = threadIdx.x;
  double db = 0.0, dg = 0.0;
  const float gam = c < a.C ? a.gamma[c] : 0.f;
  for (int n = 0; n < N; ++n) {
    double s1 = 0.0, s2 = 0.0;
    if (c < a.C)
      for (int s = 0; s < S; ++s) { const float* p = a.part + (((size_t)n * S + s) * a.C + c) * 2; s1 += (double)p[0]; s2 += (double)p[1]; }
    db += s1; dg += s2;
    g1[c] = s1 * (double)gam; g2[c] = s2 * (double)gam;
    __syncthreads();
    if (c < a.C && c % a.cpg == 0) {
      const int g = c / a.cpg;
      double t1 = 0.0, t2 = 0.0;
      for (int k = 0; k < a.cpg; ++k) { t1 += g1[c + k]; t2 += g2[c + k]; }
      const double cnt = (double)a.HW * a.cpg;
      a.coef[((size_t)n * a.G + g) * 2] = (float)(t1 / cnt);
      a.coef[((size_t)n * a.G + g) * 2 + 1] = (float)(t2 / cnt);
    }
    __syncthreads();
  }
  if (c < a.C) { a.dbeta[c] += (float)db; a.dgamma[c] += (float)dg; }
}

// dy = rstd * (dz * gamma - m1 - xhat * m2), in place over g
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnArgs a, int N) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = (size_t)N * a.HW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    const int n = (int)(v / ((size_t)a.HW * vpc));
    const int c = cv * VEC;
    float y[VEC], g[VEC];
    EV<T>::unpack(ldv<T>(a.y, v), y);
    EV<T>::unpack(ldv<T>(a.g, v), g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int gi = (c + i) / a.cpg;
      const float sc = a.ss[((size_t)n * a.C + c + i) * 2], sh = a.ss[((size_t)n * a.C + c + i) * 2 + 1];
      const float mu = a.stat[((size_t)n * a.G + gi) * 2], rs = a.stat[((size_t)n * a.G + gi) * 2 + 1];
      const float m1 = a.coef[((size_t)n * a.G + gi) * 2], m2 = a.coef[((size_t)n * a.G + gi) * 2 + 1];
      const float dz = fmaf(y[i], sc, sh) > 0.f ? g[i] : 0.f;
      const float xh = (y[i] - mu) * rs;
      g[i] = rs * (dz * a.gamma[c + i] - m1 - xh * m2);
    }
    stv<T>(a.dy, v, EV<T>::pack(g));
  }
}

// out = resample(relu(y * scale[n][c] + shift[n][c])): identity (up = 1) or bilinear x2, align_corners=True (F.interpolate in
// Conv3x3GNReLU).  The four source vectors come from L2; the activation is recomputed per tap (two FMAs + max per element).
template <typename T>
__global__ __launch_bounds__(256) void gn_act_up_kernel(const GnArgs a, int N, int H, int W, int up, float sy, float sx) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const int OH = H * up, OW = W * up;
  const size_t nvec = (size_t)N * OH * OW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    const int c = cv * VEC;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = a.ss[((size_t)n * a.C + c + i) * 2]; sh[i] = a.ss[((size_t)n * a.C + c + i) * 2 + 1]; }
    auto act = [&](int iy, int ix, float* x) {
      EV<T>::unpack(ldv<T>(a.y, (((size_t)n * H + iy) * W + ix) * vpc + cv), x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float t = fmaf(x[i], sc[i], sh[i]); x[i] = t < 0.f ? 0.f : t; }
    };
    float o[VEC];
    if (up == 1) {
      act(oy, ox, o);
    } else {
      const Lerp ly = lerp_of(oy, H, sy), lx = lerp_of(ox, W, sx);
      float a00[VEC], a01[VEC], a10[VEC], a11[VEC];
      act(ly.i0, lx.i0, a00); act(ly.i0, lx.i1, a01); act(ly.i1, lx.i0, a10); act(ly.i1, lx.i1, a11);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = ly.w0 * (lx.w0 * a00[i] + lx.w1 * a01[i]) + ly.w1 * (lx.w0 * a10[i] + lx.w1 * a11[i]);
    }
    stv<T>(a.out, v, EV<T>::pack(o));
  }
}

// Adjoint of the align_corners=True bilinear resample by `up` (2 or 4), NHWC: gin[iy][ix] = sum over the outputs whose two taps per
// axis include (iy, ix), with the very weights the forward used (same float expressions).  Gather form: one writer per element.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_adjoint_kernel(const void* gout, void* gin, int N, int H, int W, int C, int up, float sy,
                                                               float sx, float inv_sy, float inv_sx) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC;
  const int OH = H * up, OW = W * up;
  const size_t nvec = (size_t)N * H * W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int n = (int)(p / H);
    // candidate outputs: those whose source coordinate lies in (i - 1, i + 1); two guard outputs on either side absorb the float rounding
    const int oy0 = max(0, (int)((float)(iy - 1) * inv_sy) - 1), oy1 = min(OH - 1, (int)((float)(iy + 1) * inv_sy) + 2);
    const int ox0 = max(0, (int)((float)(ix - 1) * inv_sx) - 1), ox1 = min(OW - 1, (int)((float)(ix + 1) * inv_sx) + 2);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const Lerp ly = lerp_of(oy, H, sy);
      const float wy = (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);   // (i0 == i1 at the last row: both weights count)
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const Lerp lx = lerp_of(ox, W, sx);
        const float wx = (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
        if (wx == 0.f) continue;
        float g[VEC];
        EV<T>::unpack(ldv<T>(gout, (((size_t)n * OH + oy) * OW + ox) * vpc + cv), g);
        const float w = wy * wx;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w, g[i], acc[i]);
      }
    }
    stv<T>(gin, v, EV<T>::pack(acc));
  }
}

// out = (a0 + a1 + a2 + a3) * m[n][c] * mscale   (MergeBlock('add') + Dropout2d; m = keep pattern, mscale = 1 / (1 - p); nullptr = identity)
template <typename T>
__global__ __launch_bounds__(256) void merge_drop_kernel(const void* a0, const void* a1, const void* a2, const void* a3, const float* m,
                                                         float mscale, void* out, size_t HW, int C, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float x[VEC], y[VEC];
    EV<T>::unpack(ldv<T>(a0, v), x);
    EV<T>::unpack(ldv<T>(a1, v), y);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += y[i];
    EV<T>::unpack(ldv<T>(a2, v), y);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += y[i];
    EV<T>::unpack(ldv<T>(a3, v), y);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += y[i];
    if (m != nullptr) {
      const int c = (int)(v % vpc) * VEC;
      const size_t n = v / (HW * vpc);
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] *= m[n * C + c + i] * mscale;
    }
    stv<T>(out, v, EV<T>::pack(x));
  }
}
// gin = gout * m[n][c]
template <typename T>
__global__ __launch_bounds__(256) void drop_bwd_kernel(const void* gout, const float* m, float mscale, void* gin, size_t HW, int C, size_t nvec) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float x[VEC];
    EV<T>::unpack(ldv<T>(gout, v), x);
    if (m != nullptr) {
      const int c = (int)(v % vpc) * VEC;
      const size_t n = v / (HW * vpc);
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] *= m[n * C + c + i] * mscale;
    }
    stv<T>(gin, v, EV<T>::pack(x));
  }
}

// logits[n][c][Y][X] = bilinear x up (align_corners=True, nn.UpsamplingBilinear2d) of z[n][c][y][x], NCHW f32 both
__global__ __launch_bounds__(256) void bilinear_nchw_kernel(const float* z, float* out, int NC, int H, int W, int up, float sy, float sx) {
  const int OH = H * up, OW = W * up;
  const size_t n = (size_t)NC * OH * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const size_t r = i / OW;
    const int oy = (int)(r % OH);
    const size_t pc = r / OH;
    const Lerp ly = lerp_of(oy, H, sy), lx = lerp_of(ox, W, sx);
    const float* b = z + pc * H * W;
    out[i] = ly.w0 * (lx.w0 * b[ly.i0 * W + lx.i0] + lx.w1 * b[ly.i0 * W + lx.i1]) +
             ly.w1 * (lx.w0 * b[ly.i1 * W + lx.i0] + lx.w1 * b[ly.i1 * W + lx.i1]);
  }
}

// ------------------------------------------------------------------ launchers
static inline int gn_slabs(size_t HW) { size_t s = HW / 1024; return s < 1 ? 1 : (s > 64 ? 64 : (int)s); }
int gn_num_slabs(size_t HW) { return gn_slabs(HW); }

#define FPN_DISPATCH(KERNEL, grid, ...)                                                        \
  do {                                                                                          \
    if (dtype == DT_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), 0, st, __VA_ARGS__);      \
    else if (dtype == DT_F16) hipLaunchKernelGGL(KERNEL<f16_t>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)

static bool gn_shape_ok(int dtype, const GnArgs& a) {
  const int vec = dtype == DT_F32 ? 4 : 8;
  const int vpc = a.C / vec;
  return a.C % vec == 0 && a.C <= 1024 && vpc >= 1 && vpc <= 256 && 256 % vpc == 0 && a.C % a.G == 0 && a.cpg == a.C / a.G;
}

// forward: statistics (two launches) + apply / resample into `out`
hipError_t launch_gn_forward(int dtype, const GnArgs& a, int N, int H, int W, int up, hipStream_t st) {
  if (!gn_shape_ok(dtype, a) || (up != 1 && up != 2)) return hipErrorInvalidValue;
  const int S = gn_slabs(a.HW);
  if (dtype == DT_F32) hipLaunchKernelGGL((gn_reduce_kernel<float, 0>), dim3(S, N), dim3(256), 0, st, a);
  else if (dtype == DT_F16) hipLaunchKernelGGL((gn_reduce_kernel<f16_t, 0>), dim3(S, N), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, 0>), dim3(S, N), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(1024), 0, st, a, S);
  const size_t nvec = (size_t)N * H * W * up * up * (a.C / (dtype == DT_F32 ? 4 : 8));
  FPN_DISPATCH(gn_act_up_kernel, dim3(grid_for(nvec, 256)), a, N, H, W, up, lerp_scale(H, H * up), lerp_scale(W, W * up));
  return hipGetLastError();
}
// backward: a.g holds d/d(relu(gn(y))) at the resolution of y (the caller has applied the resample's adjoint); a.dy may alias a.g
hipError_t launch_gn_backward(int dtype, const GnArgs& a, int N, hipStream_t st) {
  if (!gn_shape_ok(dtype, a) || dtype == DT_F16) return hipErrorInvalidValue;
  const int S = gn_slabs(a.HW);
  if (dtype == DT_F32) hipLaunchKernelGGL((gn_reduce_kernel<float, 1>), dim3(S, N), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, 1>), dim3(S, N), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, a, S, N);
  const size_t nvec = (size_t)N * a.HW * (a.C / (dtype == DT_F32 ? 4 : 8));
  if (dtype == DT_F32) hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, a, N);
  else hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, a, N);
  return hipGetLastError();
}
hipError_t launch_bilinear_adjoint(int dtype, const void* gout, void* gin, int N, int H, int W, int C, int up, hipStream_t st) {
  if (dtype == DT_F16 || C % (dtype == DT_F32 ? 4 : 8) != 0 || up < 2) return hipErrorInvalidValue;
  const size_t nvec = (size_t)N * H * W * (C / (dtype == DT_F32 ? 4 : 8));
  const float sy = lerp_scale(H, H * up), sx = lerp_scale(W, W * up);
  const float isy = sy > 0.f ? 1.f / sy : 0.f, isx = sx > 0.f ? 1.f / sx : 0.f;
  if (dtype == DT_F32) hipLaunchKernelGGL(bilinear_adjoint_kernel<float>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, gout, gin, N, H, W, C, up, sy, sx, isy, isx);
  else hipLaunchKernelGGL(bilinear_adjoint_kernel<bf16_t>, dim3(grid_for(nvec, 256)), dim3(256), 0, st, gout, gin, N, H, W, C, up, sy, sx, isy, isx);
  return hipGetLastError();
}
hipError_t launch_merge_drop(int dtype, const void* a0, const void* a1, const void* a2, const void* a3, const float* m, float mscale, void* out,
                             int N, size_t HW, int C, hipStream_t st) {
  const size_t nvec = (size_t)N * HW * (C / (dtype == DT_F32 ? 4 : 8));
  FPN_DISPATCH(merge_drop_kernel, dim3(grid_for(nvec, 256)), a0, a1, a2, a3, m, mscale, out, HW, C, nvec);
  return hipGetLastError();
}
hipError_t launch_drop_bwd(int dtype, const void* gout, const float* m, float mscale, void* gin, int N, size_t HW, int C, hipStream_t st) {
  const size_t nvec = (size_t)N * HW * (C / (dtype == DT_F32 ? 4 : 8));
  FPN_DISPATCH(drop_bwd_kernel, dim3(grid_for(nvec, 256)), gout, m, mscale, gin, HW, C, nvec);
  return hipGetLastError();
}
hipError_t launch_bilinear_nchw(const float* z, float* out, int NC, int H, int W, int up, hipStream_t st) {
  const size_t n = (size_t)NC * H * W * up * up;
  hipLaunchKernelGGL(bilinear_nchw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, z, out, NC, H, W, up, lerp_scale(H, H * up), lerp_scale(W, W * up));
  return hipGetLastError();
}

}  // namespace octseg
