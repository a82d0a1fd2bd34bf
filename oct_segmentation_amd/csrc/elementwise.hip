// elementwise.hip -- the HBM-bound kernels of the hot path (gfx950): BatchNorm finalize /
// apply / backward, residual + skip adds, maxpool, stem im2col, Dice loss, weight packing and
// the fused optimizers.  Every tensor is NHWC with 16-byte vector accesses (8 x bf16 or 4 x f32
// per lane); per-channel reductions keep a fixed channel vector per thread so the partial sums
// stay in registers, then combine through LDS and a deterministic slab (no float atomics on the
// BN path).
#include "common.h"
#include "ev.h"

#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace octseg {

// Per-channel totals of a BN partial-sum slab ([rows][C][2] floats, one row per producing workgroup), in double, in a fixed order
// (deterministic).  A block of 1024 threads owns `cpb` consecutive channels (4..32, a power of two) and walks the rows 1024 / cpb at a time
// with eight loads in flight per thread; grid.y > 1 splits very long slabs into row groups whose partials the last block to finish (ticket)
// adds up.  Returns true in the one thread per channel (c < C) that holds the totals in s1 / s2.  The counter resets itself.
// Why channels-per-block shrink with the slab: conv_mfma's epilogue writes one row per tile, 7744..15488 rows for the 352^2 layers of
// U-Net++; with 32 channels per block a 64-channel BatchNorm was reduced by two blocks plus row groups behind a ticket whose
// __threadfence() (an L2 write-back across the XCDs) costs more than the walk -- 40 us per finalize on the forward's dependent chain,
// 5.1 ms of a 75 ms step, and MORE row groups measured slower still (512 rows per group: 76.8 vs 75.2 ms).
__device__ __forceinline__ bool slab_sum(const float* __restrict__ slab, int rows, int C, int cpb, double* part, unsigned* counters,
                                         double& s1, double& s2, int& c) {
  __shared__ double red[1280][2];
  __shared__ unsigned is_last;
  const int sh = __ffs(cpb) - 1, RL = (int)blockDim.x >> sh, pitch = cpb + 1;
  const int rl = threadIdx.x >> sh, cl = threadIdx.x & (cpb - 1);
  const int G = gridDim.y;
  c = blockIdx.x * cpb + cl;
  const int per = (rows + G - 1) / G;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  s1 = 0.0; s2 = 0.0;
  if (c < C) {
    int r = r0 + rl;
    for (; r + 7 * RL < r1; r += 8 * RL) {   // eight independent loads in flight
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const float2*)(slab + ((size_t)(r + j * RL) * C + c) * 2);
      s1 += (((double)v[0].x + (double)v[1].x) + ((double)v[2].x + (double)v[3].x)) + (((double)v[4].x + (double)v[5].x) + ((double)v[6].x + (double)v[7].x));
      s2 += (((double)v[0].y + (double)v[1].y) + ((double)v[2].y + (double)v[3].y)) + (((double)v[4].y + (double)v[5].y) + ((double)v[6].y + (double)v[7].y));
    }
    for (; r < r1; r += RL) {
      const float2 v = *(const float2*)(slab + ((size_t)r * C + c) * 2);
      s1 += (double)v.x; s2 += (double)v.y;
    }
  }
  // fold the row lanes: lanes >= 8 hand over through LDS, lanes 0..7 add every eighth, lane 0 adds those
  auto fold = [&]() {
    if (rl >= 8) { red[rl * pitch + cl][0] = s1; red[rl * pitch + cl][1] = s2; }
    __syncthreads();
    if (rl < 8) {
      for (int k = rl + 8; k < RL; k += 8) { s1 += red[k * pitch + cl][0]; s2 += red[k * pitch + cl][1]; }
      red[rl * pitch + cl][0] = s1; red[rl * pitch + cl][1] = s2;
    }
    __syncthreads();
    if (rl == 0)
      for (int k = 1; k < 8 && k < RL; ++k) { s1 += red[k * pitch + cl][0]; s2 += red[k * pitch + cl][1]; }
  };
  fold();
  const bool owner = rl == 0 && c < C;
  if (G == 1) return owner;
  if (owner) {
    part[((size_t)blockIdx.y * C + c) * 2] = s1;
    part[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(&counters[blockIdx.x], 1u);
    is_last = old == (unsigned)(G - 1);
    if (is_last) counters[blockIdx.x] = 0u;
  }
  __syncthreads();
  if (!is_last) return false;   // block-uniform
  __threadfence();
  s1 = 0.0; s2 = 0.0;
  if (c < C)
    for (int g = rl; g < G; g += RL) {
      s1 += __hip_atomic_load(&part[((size_t)g * C + c) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(&part[((size_t)g * C + c) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  fold();
  return owner;
}
// channels per block of the finalize grid: few rows -> 32 (one 256-byte segment per row), long slabs -> down to 4 (a 32-byte sector per
// row; the slab was just written and sits in L2), so that a thread walks at most a few dozen rows
static inline int slab_cpb(int rows) {
  static const int force = getenv("OCTSEG_SLAB_CPB") ? atoi(getenv("OCTSEG_SLAB_CPB")) : 0;   // A/B switch
  if (force == 4 || force == 8 || force == 16 || force == 32) return force;
  return rows <= 512 ? 32 : rows <= 2048 ? 16 : rows <= 6144 ? 8 : 4;
}
// row groups: only where a thread would still walk more than 256 rows (slabs beyond 64 k rows at 4 channels per block), within the scratch
// (G * C <= SLAB_PART_CAP partials, 64 tickets)
static inline int slab_groups(int rows, int C) {
  const int cpb = slab_cpb(rows);
  if ((C + cpb - 1) / cpb > 64) return 1;
  int g = rows / (256 * (1024 / cpb));
  const int cap = SLAB_PART_CAP / (C < 32 ? 32 : ((C + 31) / 32) * 32);
  if (g > cap) g = cap;
  if (g > 64) g = 64;
  return g < 1 ? 1 : g;
}

__global__ __launch_bounds__(1024) void bn_finalize_train_kernel(
    const float* __restrict__ slab, int rows, int C, double count, const float* gamma, const float* beta,
    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
    float* mean_out, float* rstd_out, double* part, unsigned* counters, int cpb) {
  double s1, s2;
  int c;
  if (!slab_sum(slab, rows, C, cpb, part, counters, s1, s2, c)) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}
hipError_t launch_bn_finalize_train(const float* slab, int rows, int C, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float* scale, float* shift, float* mean,
                                    float* rstd, double* part, unsigned* counters, hipStream_t st) {
  const int cpb = slab_cpb(rows);
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + cpb - 1) / cpb, slab_groups(rows, C)), dim3(1024), 0, st, slab, rows, C, count,
                     gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd, part, counters, cpb);
  return hipGetLastError();
}

// BatchNorm statistics of a SMALL tensor (<= BN_SMALL_COUNT values per channel), two-pass in double straight from the conv output:
// mean first, then the sum of squared deviations -- what torch's CPU / cuDNN kernels do.  The slab path's E[x^2] - E[x]^2 over float
// partial sums carries a relative variance error of ~2e-7 (1 + mean^2 / var): harmless on feature maps with mean ~ std, but 1 % of rstd
// on the B nearly equal numbers of DeepLabV3+'s pooled ASPP branch, and visible (1e-5 of rstd, 1e-2 of small gradients) behind the +-8
// BatchNorm biases of the kink-free parity nets on 2x2 .. 8x8 maps.  Small tensors cost nothing to read twice (they sit in L2); the
// 704^2 workloads at >= 4 frames per GPU never take this path (16 x 22 x 22 = 7744 values per channel in their smallest BatchNorm).
// Block = 32 channels x 32 row lanes, four loads in flight per thread; same outputs as bn_finalize_train_kernel.
// (BN_SMALL_COUNT = 1024: at 4096 with 8 row lanes the walk was ~100 us per BatchNorm and the 2 x 44 x 44 = 3872-value layer3 BatchNorms
// of a 2-frame U-Net++/resnet101 step took it 69 times: 17.5 -> 30.9 ms per step, measured.)
template <typename T>
__global__ __launch_bounds__(1024) void bn_finalize_small_kernel(const void* y, int count, int C, const float* gamma, const float* beta,
                                                                 float* running_mean, float* running_var, float momentum, float eps,
                                                                 float* scale, float* shift, float* mean_out, float* rstd_out) {
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < C;
  auto val = [&](int i) -> double {
    if (sizeof(T) == 4) return (double)((const float*)y)[(size_t)i * C + c];
    return (double)__uint_as_float((unsigned)((const unsigned short*)y)[(size_t)i * C + c] << 16);
  };
  auto fold = [&](double x) -> double {   // sum over the 32 row lanes, in a fixed order, to every lane
    __syncthreads();
    red[rl][cl] = x;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < 32; ++k) t += red[k][cl];
    return t;
  };
  double s = 0.0;
  if (live) {
    int i = rl;
    for (; i + 96 < count; i += 128) { const double a0 = val(i), a1 = val(i + 32), a2 = val(i + 64), a3 = val(i + 96); s += (a0 + a1) + (a2 + a3); }
    for (; i < count; i += 32) s += val(i);
  }
  const double m = fold(s) / (double)count;
  double v = 0.0;
  if (live) {
    int i = rl;
    for (; i + 96 < count; i += 128) {
      const double d0 = val(i) - m, d1 = val(i + 32) - m, d2 = val(i + 64) - m, d3 = val(i + 96) - m;
      v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (; i < count; i += 32) { const double d = val(i) - m; v += d * d; }
  }
  v = fold(v);
  if (!live || rl != 0) return;
  const double var = v / (double)count;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  mean_out[c] = (float)m;
  rstd_out[c] = rstd;
  const double unbiased = count > 1 ? v / (double)(count - 1) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}
hipError_t launch_bn_finalize_small(int dtype, const void* y, int count, int C, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                                    hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const dim3 grid((C + 31) / 32);
  if (dtype == DT_F32) hipLaunchKernelGGL(bn_finalize_small_kernel<float>, grid, dim3(1024), 0, st, y, count, C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
  else hipLaunchKernelGGL(bn_finalize_small_kernel<bf16_t>, grid, dim3(1024), 0, st, y, count, C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, rstd);
  return hipGetLastError();
}

__global__ void bn_finalize_eval_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                        const float* rv, float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}
// every BatchNorm of a plan in one launch (eval): job j covers channels [prefix[j], prefix[j+1])
__global__ __launch_bounds__(256) void bn_finalize_eval_all_kernel(const float* params, const float* buffers, char* ws, const BnEvalJob* tab,
                                                                   const unsigned* prefix, int njobs, unsigned total, float eps) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= g) lo = mid; else hi = mid - 1;
  }
  const BnEvalJob jb = tab[lo];
  const int c = (int)(g - prefix[lo]);
  const float sc = params[jb.gamma_off + c] / sqrtf(buffers[jb.rv_off + c] + (jb.eps > 0.f ? jb.eps : eps));
  float* ss = (float*)(ws + jb.ss_off);
  ss[c] = sc;
  float sh = params[jb.beta_off + c] - buffers[jb.rm_off + c] * sc;
  if (jb.bias_off != ~(size_t)0) sh = fmaf(params[jb.bias_off + c], sc, sh);   // (conv + b) * scale + shift: ConvTranspose2d of LinkNet
  ss[jb.C + c] = sh;
}
hipError_t launch_bn_finalize_eval_all(const float* params, const float* buffers, void* ws, const BnEvalJob* tab, const unsigned* prefix,
                                       int njobs, unsigned total, float eps, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_eval_all_kernel, dim3((total + 255) / 256), dim3(256), 0, st, params, buffers, (char*)ws, tab, prefix,
                     njobs, total, eps);
  return hipGetLastError();
}
hipError_t launch_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift,
                                   hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  return hipGetLastError();
}

// ------------------------------------------------------------------ BN apply (+residual, +relu, +skip)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const BnActArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const size_t nvec = a.npix * (size_t)(a.C / VEC);
  const int vpc = a.C / VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % vpc) * VEC;
    float x[VEC];
    EV<T>::unpack(ldv<T>(a.y, v), x);
    if (a.scale) {   // (nullptr: BatchNorm already folded into the producing conv, eval)
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = fmaf(x[i], a.scale[c + i], a.shift[c + i]);
    }
    if (a.res) {
      float rr[VEC];
      EV<T>::unpack(ldv<T>(a.res, v), rr);
      if (a.rscale) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) rr[i] = fmaf(rr[i], a.rscale[c + i], a.rshift[c + i]);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] += rr[i];
    }
    if (a.relu) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = x[i] < 0.f ? 0.f : x[i];   // (torch's relu keeps a NaN; fmaxf would turn it into 0)
    }
    if (a.post) {
      float pp[VEC];
      EV<T>::unpack(ldv<T>(a.post, v), pp);
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] += pp[i];
    }
    if (a.maskbits != nullptr) {
      unsigned bits = 0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) bits |= (x[i] > 0.f ? 1u : 0u) << i;
      a.maskbits[v] = (unsigned char)bits;
    }
    stv<T>(a.out, v, EV<T>::pack(x));
  }
}
// grid for the per-channel elementwise kernels: the stride (grid * 256 threads) must be a multiple of vpc
static inline int grid_for_channels(size_t nvec, int vpc) {
  int g = grid_for(nvec, 256);
  int a = vpc, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }   // a = gcd(vpc, 256)
  const int m = vpc / a;                              // the grid must be a multiple of m
  g = (g / m) * m;
  return g < m ? m : g;
}
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t st) {
  const int vpc = a.C / (dtype == DT_F32 ? 4 : 8);
  const size_t nvec = a.npix * (size_t)vpc;
  const int g = grid_for_channels(nvec, vpc);
  if (dtype == DT_F16) hipLaunchKernelGGL(bn_act_kernel<f16_t>, dim3(g), dim3(256), 0, st, a);
  else if (dtype == DT_F32) hipLaunchKernelGGL(bn_act_kernel<float>, dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(g), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ BN backward
// dz = g * mask;  slab[row][c] = (sum dz, sum dz * xhat),  xhat = (y - mean) * rstd
// FUSED: the kernel also finishes the reduction (what bn_bwd_finalize_kernel did in a launch of its own, 126 per step): rows are
// grouped by 32; the workgroup whose ticket completes a group sums that group's rows in row order (double) into a partial, the one
// whose ticket completes the last group sums the <= 32 partials in group order and writes dgamma / dbeta / coef -- a fixed summation
// order, so still deterministic.  Hand-off (cdna_hip_programming.md Guideline 16 / split-K recipe): slab rows and partials are stored
// write-through (sc1: agent-scope relaxed atomic stores), every storing wave drains vmcnt, workgroup barrier, ONE lane takes the ticket
// (relaxed agent fetch_add); the finisher does one agent acquire (+ vmcnt drain + barrier) and then reads with plain loads, all in flight
// at once (sc1 / atomic loads were issued one by one: 64 dependent round trips per launch cost 60 us, measured).
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ float red[256][VEC * 2 + 1];
  __shared__ unsigned s_last;
  const int vpc = a.C / VEC;                     // vectors per pixel (power of two)
  const int tpv = vpc >= 256 ? 1 : 256 / vpc;    // threads sharing one channel vector
  const int cv = (vpc >= 256 ? blockIdx.y * 256 : 0) + (threadIdx.x % (vpc >= 256 ? 256 : vpc));
  const int pl = vpc >= 256 ? 0 : threadIdx.x / vpc;
  const int c = cv * VEC;
  float sc[VEC], sh[VEC], mu[VEC], rs[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sc[i] = a.scale[c + i]; sh[i] = a.shift[c + i]; mu[i] = a.mean[c + i]; rs[i] = a.rstd[c + i]; }
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  for (size_t p = (size_t)blockIdx.x * tpv + pl; p < a.npix; p += (size_t)gridDim.x * tpv) {
    const size_t v = p * vpc + cv;
    float g[VEC], y[VEC];
    EV<T>::unpack(ldv<T>(a.g, v), g);
    EV<T>::unpack(ldv<T>(a.y, v), y);
    if (a.mask == 2) {
      if (a.maskbits != nullptr) {
        const unsigned bits = a.maskbits[v];
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (!((bits >> i) & 1u)) g[i] = 0.f;
      } else {
        float o[VEC];
        EV<T>::unpack(ldv<T>(a.out, v), o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (!(o[i] > 0.f)) g[i] = 0.f;
      }
    } else if (a.mask == 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) if (!(fmaf(y[i], sc[i], sh[i]) > 0.f)) g[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s1[i] += g[i]; s2[i] += g[i] * (y[i] - mu[i]) * rs[i]; }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[threadIdx.x][i] = s1[i]; red[threadIdx.x][VEC + i] = s2[i]; }
  __syncthreads();
  if (pl == 0) {
    for (int k = 1; k < tpv; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s1[i] += red[threadIdx.x + k * vpc][i]; s2[i] += red[threadIdx.x + k * vpc][VEC + i]; }
    float* o = a.slab + ((size_t)blockIdx.x * a.C + c) * 2;
    if constexpr (FUSED) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {   // one 8-byte write-through store per channel: (sum dz, sum dz * xhat)
        const unsigned long long bits = (unsigned long long)__float_as_uint(s1[i]) | ((unsigned long long)__float_as_uint(s2[i]) << 32);
        __hip_atomic_store((unsigned long long*)(o + 2 * i), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { o[2 * i] = s1[i]; o[2 * i + 1] = s2[i]; }
    }
  }
  if constexpr (FUSED) {
    const int rows = gridDim.x, G = (rows + 31) / 32, grp = blockIdx.x / 32;
    const int r0 = grp * 32, r1 = min(rows, r0 + 32);
    const int cb0 = (vpc >= 256 ? blockIdx.y * 256 : 0) * VEC;            // first channel of this block column
    const int nval = (vpc >= 256 ? 256 : vpc) * VEC * 2;                    // (sum, sum) pairs of its channels, as floats
    unsigned* cg = a.fcnt + blockIdx.y * 33 * 32;                           // tickets of this column: group g at [g * 32] (a 128-byte line each: atomics on
                                                                            // one line serialise at ~12 ns), the top ticket at [32 * 32]
    double* fp = a.fpart + (size_t)blockIdx.y * 32 * (256 * VEC * 2);       // [group][nval] partials of this column
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every storing wave: its row is out
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(&cg[grp * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)(r1 - r0 - 1) ? 1u : 0u;
      if (s_last) {
        cg[grp * 32] = 0u;                                                  // ready for the next launch (kernel boundary publishes it)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!s_last) return;
    // (plain loads behind the acquire: 32 independent loads per value in flight, summed in row order)
    for (int i = threadIdx.x; i < nval; i += 256) {
      const float* col = a.slab + ((size_t)r0 * a.C + cb0) * 2 + i;
      float v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) v[k] = col[(size_t)min(k, r1 - r0 - 1) * a.C * 2];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) s += k < r1 - r0 ? (double)v[k] : 0.0;
      __hip_atomic_store(fp + (size_t)grp * nval + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(&cg[32 * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)(G - 1) ? 1u : 0u;
      if (s_last) {
        cg[32 * 32] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < nval; i += 256) {
      double v[32];
#pragma unroll
      for (int g = 0; g < 32; ++g) v[g] = fp[(size_t)min(g, G - 1) * nval + i];
      double s = 0.0;
#pragma unroll
      for (int g = 0; g < 32; ++g) s += g < G ? v[g] : 0.0;
      const int ch = cb0 + (i >> 1);
      if (i & 1) { a.dgamma[ch] += (float)s; a.coef[2 * ch + 1] = (float)(s / (double)a.npix); }
      else { a.dbeta[ch] += (float)s; a.coef[2 * ch] = (float)(s / (double)a.npix); }
    }
  }
}
// MEASURED (round 3, U-Net++/resnet101, one box): slower than the separate finalize launch it replaces -- BatchNorm sweeps alone 12.87 ->
// 15.67 ms per step at 16 frames, 3.36 -> 5.94 at 2 frames (+22 us per launch: two agent acquires at 4-8 resident workgroups per CU cost
// ~7 us each, MI355X_MICROARCH.md fence table, plus the ticket round trips; the launch boundary it saves is ~12 us).  Opt-in only.
bool bn_bwd_fused_finalize() {
  static const bool on = getenv("OCTSEG_FUSED_BNFIN") != nullptr;   // A/B switch (default: separate bn_bwd_finalize launch)
  return on;
}
hipError_t launch_bn_bwd_reduce(int dtype, const BnBwdArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int VEC = dtype == DT_F32 ? 4 : 8;
  const int vpc = a.C / VEC;
  dim3 grid(a.rows, vpc >= 256 ? vpc / 256 : 1);
  const bool fused = a.fpart != nullptr && a.fcnt != nullptr;
  if (fused && (grid.y > 8 || a.rows > 1024)) return hipErrorInvalidValue;
  if (dtype == DT_F32) {
    if (fused) hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, false>), grid, dim3(256), 0, st, a);
  } else {
    if (fused) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, false>), grid, dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// Small tensors (<= BN_SMALL_COUNT values per channel): reduction, finalize AND apply in one launch, all in double as torch's CPU
// kernels do (acc_type: sums, k = dotp * invstd^2 / n and dx = (dy - mean_dy - (x - mean) k) invstd w are evaluated in double and
// rounded once).  sum dz * xhat is a sum of cancelling terms and the apply pass another cancellation; in float the engine sat 80..500x
// further from the float64 gradient than torch on 2x2 .. 8x8 maps (1e-2 of a small gradient; found by tests/test_gpu_fuzz_f4.py).
// Block = 32 channels x 32 row lanes; a thread reads and writes its own rows only (dy may alias g); same masking and identity-shortcut
// hand-off as bn_bwd_reduce_kernel / bn_bwd_apply_kernel.
template <typename T>
__global__ __launch_bounds__(1024) void bn_bwd_small_kernel(const BnBwdArgs a) {
  constexpr int VEC = EV<T>::VEC;
  __shared__ double red[32][33][2];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < a.C;
  auto val = [&](const void* p, size_t i) -> float {
    if (sizeof(T) == 4) return ((const float*)p)[i];
    return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16);
  };
  auto put = [&](void* p, size_t i, float v) {
    if (sizeof(T) == 4) ((float*)p)[i] = v;
    else ((unsigned short*)p)[i] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
  };
  const float sc = live ? a.scale[c] : 0.f, sh = live ? a.shift[c] : 0.f;
  const double mu = live ? (double)a.mean[c] : 0.0, rs = live ? (double)a.rstd[c] : 0.0;
  const int vpc = a.C / VEC, cv = c / VEC, ci = c - cv * VEC;
  auto masked_g = [&](size_t p, size_t e, float y) -> float {
    float g = val(a.g, e);
    if (a.mask == 2) {
      if (a.maskbits != nullptr) { if (!((a.maskbits[p * vpc + cv] >> ci) & 1u)) g = 0.f; }
      else if (!(val(a.out, e) > 0.f)) g = 0.f;
    } else if (a.mask == 1) {
      if (!(fmaf(y, sc, sh) > 0.f)) g = 0.f;
    }
    return g;
  };
  double s1 = 0.0, s2 = 0.0;
  if (live)
    for (size_t p = rl; p < a.npix; p += 32) {
      const size_t e = p * a.C + c;
      const float y = val(a.y, e);
      const float g = masked_g(p, e, y);
      s1 += (double)g;
      s2 += (double)g * (((double)y - mu) * rs);
    }
  red[rl][cl][0] = s1; red[rl][cl][1] = s2;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  for (int k = 0; k < 32; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }   // every lane: the same fixed order
  if (!live) return;
  if (rl == 0) {
    a.dbeta[c] += (float)s1;
    a.dgamma[c] += (float)s2;
    a.coef[2 * c] = (float)(s1 / (double)a.npix);
    a.coef[2 * c + 1] = (float)(s2 / (double)a.npix);
  }
  const double c0 = s1 / (double)a.npix, c1 = s2 / (double)a.npix, A = (double)a.gamma[c] * rs;
  for (size_t p = rl; p < a.npix; p += 32) {
    const size_t e = p * a.C + c;
    const float y = val(a.y, e);
    const float g = masked_g(p, e, y);
    if (a.res_grad != nullptr) put(a.res_grad, e, a.res_store ? g : g + val(a.res_grad, e));
    put(a.dy, e, (float)(A * ((double)g - c0 - (((double)y - mu) * rs) * c1)));
  }
}
hipError_t launch_bn_bwd_small(int dtype, const BnBwdArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const dim3 grid((a.C + 31) / 32);
  if (dtype == DT_F32) hipLaunchKernelGGL(bn_bwd_small_kernel<float>, grid, dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(bn_bwd_small_kernel<bf16_t>, grid, dim3(1024), 0, st, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const BnBwdArgs a, const int cpb) {
  double s1, s2;
  int c;
  if (!slab_sum(a.slab, a.rows, a.C, cpb, a.part, a.counters, s1, s2, c)) return;
  a.dbeta[c] += (float)s1;
  a.dgamma[c] += (float)s2;
  a.coef[2 * c] = (float)(s1 / (double)a.npix);
  a.coef[2 * c + 1] = (float)(s2 / (double)a.npix);
}
hipError_t launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t st) {
  const int cpb = slab_cpb(a.rows);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + cpb - 1) / cpb, slab_groups(a.rows, a.C)), dim3(1024), 0, st, a, cpb);
  return hipGetLastError();
}

// dy = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
// MM = how the ReLU mask is found (0 none, 1 recomputed from y, 2 from the stored activation, 3 from bn_act's bit planes): a template parameter
// so that each variant only keeps its own operands in registers, and at most 128 of them (four waves per SIMD, launch bound): the weight-gradient
// kernels on the side stream hold 380-384 of a SIMD's 512 registers per lane for a whole launch, and a sweep wave that needs more than the
// remaining 128 cannot start beside them -- the sweep then waits for compute units instead of hiding under the matrix work.  (Eight waves per
// SIMD -- 64 registers, two vectors per thread in flight -- spills 28-52 registers in the bf16 variants: 11.7 -> 26.3 ms of sweeps per step.)
template <typename T, int MM>
__global__ __launch_bounds__(256, 4) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = a.C / VEC;
  const size_t nvec = a.npix * (size_t)vpc;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t v0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(v0 % vpc) * VEC;   // loop invariant: the stride is a multiple of vpc (launch_bn_bwd_apply)
  // dy = A * dz - (A * c1 - A * c2 * mean * rstd) - (A * c2 * rstd) * y,  A = gamma * rstd: three FMAs per element
  float sc[VEC], sh[VEC], mu[VEC], rs[VEC], A[VEC], c1[VEC], c2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sc[i] = a.scale[c + i]; sh[i] = a.shift[c + i]; mu[i] = a.mean[c + i]; rs[i] = a.rstd[c + i];
    A[i] = a.gamma[c + i] * rs[i]; c1[i] = a.coef[2 * (c + i)]; c2[i] = a.coef[2 * (c + i) + 1];
  }
  // U vectors per thread and sweep, every load issued before the first use: one 16-byte load per operand and thread
  // in flight reached only ~2.4 TB/s on these three-stream sweeps
  constexpr int U = 4;
  for (size_t vb = v0; vb < nvec; vb += U * stride) {
    uint4 gv[U], yv[U], ov[U];
    unsigned mb[U];
    constexpr bool use_bits = MM == 3;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t v = vb + u * stride < nvec ? vb + u * stride : vb;
      gv[u] = ldv<T>(a.g, v); yv[u] = ldv<T>(a.y, v);
      mb[u] = 0;
      if (use_bits) mb[u] = a.maskbits[v];
      else if (MM == 2) ov[u] = ldv<T>(a.out, v);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t v = vb + u * stride;
      if (v >= nvec) break;
      float g[VEC], y[VEC];
      EV<T>::unpack(gv[u], g);
      EV<T>::unpack(yv[u], y);
      if (use_bits) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (!((mb[u] >> i) & 1u)) g[i] = 0.f;
      } else if (MM == 2) {
        float o[VEC];
        EV<T>::unpack(ov[u], o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (!(o[i] > 0.f)) g[i] = 0.f;
      } else if (MM == 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (!(fmaf(y[i], sc[i], sh[i]) > 0.f)) g[i] = 0.f;
      }
      if (a.res_grad != nullptr) {   // the masked gradient also flows into the block's identity shortcut (masked_accum's arithmetic)
        float d[VEC];
        if (a.res_store) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) d[i] = g[i];
        } else {
          EV<T>::unpack(ldv<T>(a.res_grad, v), d);
#pragma unroll
          for (int i = 0; i < VEC; ++i) d[i] += g[i];
        }
        stv<T>(a.res_grad, v, EV<T>::pack(d));
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xh = (y[i] - mu[i]) * rs[i];
        g[i] = A[i] * (g[i] - c1[i] - xh * c2[i]);
      }
      stv<T>(a.dy, v, EV<T>::pack(g));
    }
  }
}
hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vpc = a.C / (dtype == DT_F32 ? 4 : 8);
  const size_t nvec = a.npix * (size_t)vpc;
  const int g = grid_for_channels(nvec, vpc);
  const int mm = a.mask == 2 ? (a.maskbits != nullptr ? 3 : 2) : a.mask;
#define OCTSEG_APPLY(MM_)                                                                                             \
  if (mm == MM_) {                                                                                                    \
    if (dtype == DT_F32) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, MM_>), dim3(g), dim3(256), 0, st, a);         \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, MM_>), dim3(g), dim3(256), 0, st, a);                        \
  }
  OCTSEG_APPLY(0) OCTSEG_APPLY(1) OCTSEG_APPLY(2) OCTSEG_APPLY(3)
#undef OCTSEG_APPLY
  return hipGetLastError();
}

// ------------------------------------------------------------------ gradient plumbing
template <typename T>
__global__ __launch_bounds__(256) void masked_accum_kernel(void* dst, const void* g, const void* om, size_t nvec, int store) {
  constexpr int VEC = EV<T>::VEC;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float d[VEC], x[VEC];
    if (store) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) d[i] = 0.f;
    } else {
      EV<T>::unpack(ldv<T>(dst, v), d);
    }
    EV<T>::unpack(ldv<T>(g, v), x);
    if (om) {
      float o[VEC];
      EV<T>::unpack(ldv<T>(om, v), o);
#pragma unroll
      for (int i = 0; i < VEC; ++i) if (!(o[i] > 0.f)) x[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] += x[i];
    stv<T>(dst, v, EV<T>::pack(d));
  }
}
hipError_t launch_masked_accum(int dtype, void* dst, const void* g, const void* out_mask, size_t n, int store, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const size_t nvec = n / (dtype == DT_F32 ? 4 : 8);
  const int gr = grid_for(nvec, 256);
  if (dtype == DT_F32) hipLaunchKernelGGL(masked_accum_kernel<float>, dim3(gr), dim3(256), 0, st, dst, g, out_mask, nvec, store);
  else hipLaunchKernelGGL(masked_accum_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, dst, g, out_mask, nvec, store);
  return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void pool2x2_accum_kernel(void* dst, const void* src, int N, int H, int W, int C, int store) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC;
  const size_t nvec = (size_t)N * H * W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    float d[VEC];
    if (store) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) d[i] = 0.f;
    } else {
      EV<T>::unpack(ldv<T>(dst, v), d);
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float s[VEC];
        const size_t sv = (((size_t)n * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * vpc + cv;
        EV<T>::unpack(ldv<T>(src, sv), s);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] += s[i];
      }
    stv<T>(dst, v, EV<T>::pack(d));
  }
}
hipError_t launch_pool2x2_accum(int dtype, void* dst, const void* src, int N, int H, int W, int C, int store, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const size_t nvec = (size_t)N * H * W * (C / (dtype == DT_F32 ? 4 : 8));
  const int gr = grid_for(nvec, 256);
  if (dtype == DT_F32) hipLaunchKernelGGL(pool2x2_accum_kernel<float>, dim3(gr), dim3(256), 0, st, dst, src, N, H, W, C, store);
  else hipLaunchKernelGGL(pool2x2_accum_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, dst, src, N, H, W, C, store);
  return hipGetLastError();
}

// bias gradient: out[c] += sum_pixels g[pixel][c]   (few channels; one atomic per block and channel)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const void* g, size_t npix, int Cstride, int C, float* out) {
  __shared__ float red[256];
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
      if (sizeof(T) == 4) s += ((const float*)g)[p * Cstride + c];
      else s += __uint_as_float((unsigned)((const bf16_t*)g)[p * Cstride + c] << 16);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + c, red[0]);
    __syncthreads();
  }
}
// The same for channel counts that fill 16-byte vectors (LinkNet's ConvTranspose2d biases, FPN's lateral convs: 16..512 channels over up
// to 16 x 352^2 pixels -- the scalar kernel above walked them channel by channel with stride-C loads: 0.9 ms per launch, 5.3 ms of a
// 31 ms LinkNet/resnet50 step).  Thread (r, v) owns channel vector v of the pixels r, r + rows, ...: coalesced 16-byte loads, register
// sums, one LDS pass over the rows, one atomic per block and channel.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_vec_kernel(const void* g, size_t npix, int Cstride, int C, int rows, float* out) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float csum[];   // [rows][C]
  const int nvc = C / VW;
  const int r = threadIdx.x / nvc, v = threadIdx.x - r * nvc;
  float s[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) s[j] = 0.f;
  if (r < rows) {
    const char* base = (const char*)g + (size_t)v * 16;
    for (size_t p = (size_t)blockIdx.x * rows + r; p < npix; p += (size_t)gridDim.x * rows) {
      const uint4 q = *(const uint4*)(base + p * (size_t)Cstride * sizeof(T));
      if (sizeof(T) == 4) {
        s[0] += __uint_as_float(q.x); s[1] += __uint_as_float(q.y); s[2] += __uint_as_float(q.z); s[3] += __uint_as_float(q.w);
      } else {
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[2 * j] += __uint_as_float(w[j] << 16); s[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u); }
      }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) csum[(size_t)r * C + v * VW + j] = s[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int k = 0; k < rows; ++k) t += csum[(size_t)k * C + c];
    atomicAdd(out + c, t);
  }
}
hipError_t launch_channel_sum(int dtype, const void* g, size_t npix, int Cstride, int C, float* out, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const int vw = dtype == DT_F32 ? 4 : 8;
  if (C % vw == 0 && Cstride % vw == 0 && C / vw <= 256 && ((uintptr_t)g & 15) == 0) {
    const int nvc = C / vw, rows = 256 / nvc;
    size_t want = (npix + (size_t)rows * 8 - 1) / ((size_t)rows * 8);
    const int gr = deterministic_mode() ? 1 : (int)(want < 1 ? 1 : want > 1024 ? 1024 : want);
    const size_t lds = (size_t)rows * C * sizeof(float);
    if (dtype == DT_F32) hipLaunchKernelGGL(channel_sum_vec_kernel<float>, dim3(gr), dim3(256), lds, st, g, npix, Cstride, C, rows, out);
    else hipLaunchKernelGGL(channel_sum_vec_kernel<bf16_t>, dim3(gr), dim3(256), lds, st, g, npix, Cstride, C, rows, out);
    return hipGetLastError();
  }
  const int gr = deterministic_mode() ? 1 : grid_for(npix, 256, 512);
  if (dtype == DT_F32) hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(gr), dim3(256), 0, st, g, npix, Cstride, C, out);
  else hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, g, npix, Cstride, C, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------ maxpool 3x3 s2 p1
// Ties resolve to the first maximum in (row, column) scan order, as torch's CPU kernel does.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* in, void* out, unsigned char* idx, int N, int H, int W, int C) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC, OH = H / 2, OW = W / 2;
  const size_t nvec = (size_t)N * OH * OW * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    float m[VEC];
    unsigned char am[VEC];   // window position (r * 3 + s) of the first maximum: the backward reads it back
#pragma unroll
    for (int i = 0; i < VEC; ++i) { m[i] = -INFINITY; am[i] = 255; }
    for (int r = 0; r < 3; ++r) {
      const int iy = 2 * oy - 1 + r;
      if (iy < 0 || iy >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int ix = 2 * ox - 1 + s;
        if (ix < 0 || ix >= W) continue;
        float x[VEC];
        EV<T>::unpack(ldv<T>(in, (((size_t)n * H + iy) * W + ix) * vpc + cv), x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (x[i] > m[i] || x[i] != x[i]) { m[i] = x[i]; am[i] = (unsigned char)(r * 3 + s); }
      }
    }
    stv<T>(out, v, EV<T>::pack(m));
    if (idx != nullptr) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) idx[v * VEC + i] = am[i];
    }
  }
}
hipError_t launch_maxpool_fwd(int dtype, const void* in, void* out, unsigned char* idx, int N, int H, int W, int C, hipStream_t st) {
  const size_t nvec = (size_t)N * (H / 2) * (W / 2) * (C / (dtype == DT_F32 ? 4 : 8));
  const int gr = grid_for(nvec, 256);
  if (dtype == DT_F16) hipLaunchKernelGGL(maxpool_fwd_kernel<f16_t>, dim3(gr), dim3(256), 0, st, in, out, idx, N, H, W, C);
  else if (dtype == DT_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(gr), dim3(256), 0, st, in, out, idx, N, H, W, C);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, in, out, idx, N, H, W, C);
  return hipGetLastError();
}

// backward from the saved window positions: an input pixel sits at position (iy - 2 oy + 1) * 3 + (ix - 2 ox + 1) of each of
// the <= 4 windows that cover it and receives that window's gradient iff the forward recorded that position
// (4 x (1 + 2) bytes per element instead of re-deriving four 9-element argmaxes)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned char* idx, const void* gout, void* gin, int N, int H, int W,
                                                              int C, int store) {
  constexpr int VEC = EV<T>::VEC;
  const int vpc = C / VEC, OH = H / 2, OW = W / 2;
  const size_t nvec = (size_t)N * H * W * vpc;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(v % vpc);
    size_t p = v / vpc;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int n = (int)(p / H);
    float gi[VEC];
    if (store) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gi[i] = 0.f;
    } else {
      EV<T>::unpack(ldv<T>(gin, v), gi);
    }
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox >= OW) continue;
        const int code = (iy - 2 * oy + 1) * 3 + (ix - 2 * ox + 1);
        const size_t ov = (((size_t)n * OH + oy) * OW + ox) * vpc + cv;
        float g[VEC];
        EV<T>::unpack(ldv<T>(gout, ov), g);
        unsigned char am[VEC];
        if (VEC == 8) { const uint2 w = *(const uint2*)(idx + ov * 8); *(uint2*)am = w; }
        else { const unsigned w = *(const unsigned*)(idx + ov * 4); *(unsigned*)am = w; }
#pragma unroll
        for (int i = 0; i < VEC; ++i) if (am[i] == code) acc[i] += g[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) gi[i] += acc[i];
    stv<T>(gin, v, EV<T>::pack(gi));
  }
}
hipError_t launch_maxpool_bwd_idx(int dtype, const unsigned char* idx, const void* gout, void* gin, int N, int H, int W, int C, int store,
                                  hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  const size_t nvec = (size_t)N * H * W * (C / (dtype == DT_F32 ? 4 : 8));
  const int gr = grid_for(nvec, 256);
  if (dtype == DT_F32) hipLaunchKernelGGL(maxpool_bwd_idx_kernel<float>, dim3(gr), dim3(256), 0, st, idx, gout, gin, N, H, W, C, store);
  else hipLaunchKernelGGL(maxpool_bwd_idx_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, idx, gout, gin, N, H, W, C, store);
  return hipGetLastError();
}

// ------------------------------------------------------------------ stem im2col (7x7 s2 p3, 3 channels)
// col[n][oy][ox][k], k = (r*7+s)*3+ci for k < 147, zero padded to KP; the (x-mean)/std normalisation
// of the reference's forward() is fused here (raw 0..255 BGR input).
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* img, void* col, int N, int H, int W, int KP,
                                                          float m0, float m1, float m2, float i0, float i1, float i2, int KS, int PT) {
  constexpr int VEC = EV<T>::VEC;
  const int OH = H / 2, OW = W / 2, vpr = KP / VEC;
  const size_t nvec = (size_t)N * OH * OW * vpr;
  const float mean[3] = {m0, m1, m2}, inv[3] = {i0, i1, i2};
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int kv = (int)(v % vpr);
    size_t p = v / vpr;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    float x[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int k = kv * VEC + i;
      float val = 0.f;
      if (k < KS * KS * 3) {   // KS = 7: 147 of KP = 160; KS = 3 (timm RegNet stem): 27 of 32
        const int tap = k / 3, ci = k - tap * 3;
        const int r = tap / KS, s = tap - r * KS;
        const int iy = 2 * oy - PT + r, ix = 2 * ox - PT + s;     // (PT: 3 ResNet, 1 RegNet, 0 EfficientNet's static 'same' padding)
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
          val = (img[(((size_t)n * 3 + ci) * H + iy) * W + ix] - mean[ci]) * inv[ci];
      }
      x[i] = val;
    }
    stv<T>(col, v, EV<T>::pack(x));
  }
}
hipError_t launch_stem_im2col(int dtype, const float* img, void* col, int N, int H, int W, int KP,
                              const float* mean, const float* stdv, int normalize, hipStream_t st, int ksize, int pad) {
  if (pad < 0) pad = ksize / 2;
  const size_t nvec = (size_t)N * (H / 2) * (W / 2) * (KP / (dtype == DT_F32 ? 4 : 8));
  const int gr = grid_for(nvec, 256);
  float m[3] = {0, 0, 0}, iv[3] = {1, 1, 1};
  if (normalize) for (int i = 0; i < 3; ++i) { m[i] = mean[i]; iv[i] = 1.0f / stdv[i]; }
  if (dtype == DT_F16)
    hipLaunchKernelGGL(stem_im2col_kernel<f16_t>, dim3(gr), dim3(256), 0, st, img, col, N, H, W, KP, m[0], m[1], m[2], iv[0], iv[1], iv[2], ksize, pad);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(gr), dim3(256), 0, st, img, col, N, H, W, KP, m[0], m[1], m[2], iv[0], iv[1], iv[2], ksize, pad);
  else
    hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, img, col, N, H, W, KP, m[0], m[1], m[2], iv[0], iv[1], iv[2], ksize, pad);
  return hipGetLastError();
}

// ------------------------------------------------------------------ Dice loss
static __device__ __forceinline__ float sigmoid_f(float z) { return 1.0f / (1.0f + __expf(-z)); }
static __device__ __forceinline__ float sigmoid_acc(float z) {
  // exp(logsigmoid(z)) as smp computes it, evaluated without cancellation
  const float e = expf(-fabsf(z));
  return z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}
template <typename V> static __device__ __forceinline__ V block_sum(V v, V* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  V s = 0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}
// grid: (chunks, C, B)
__global__ __launch_bounds__(256) void dice_fwd_kernel(const DiceArgs a) {
  __shared__ double dred[4];
  __shared__ long long ired[4];
  const int c = blockIdx.y, b = blockIdx.z;
  const float* z = a.logits + ((size_t)b * a.C + c) * a.HW;
  const float* t = a.target + ((size_t)b * a.C + c) * a.HW;
  double sI = 0, sS = 0, sT = 0, sB = 0;
  long long tp = 0, np = 0, nt = 0;
  const bool want_bce = a.loss_kind != LOSS_DICE;
  auto one = [&](float zi, float ti) {
    const float p = sigmoid_acc(zi);
    sI += (double)(p * ti); sS += (double)(p + ti); sT += (double)ti;
    // binary_cross_entropy_with_logits, ATen's stable form: (1 - t) z + max(-z, 0) + log(1 + exp(-|z|))
    if (want_bce) sB += (double)((1.0f - ti) * zi + fmaxf(-zi, 0.f) + log1pf(expf(-fabsf(zi))));
    const int pred = p > 0.5f;
    const int tt = (long long)ti != 0;  // .long() truncation as get_stats does
    tp += pred & tt; np += pred; nt += tt;
  };
  if ((a.HW & 3) == 0) {   // 16-byte loads (a plane starts on a 16-byte boundary then)
    const float4* z4 = (const float4*)z; const float4* t4 = (const float4*)t;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HW / 4; i += (size_t)gridDim.x * blockDim.x) {
      const float4 zv = z4[i], tv = t4[i];
      one(zv.x, tv.x); one(zv.y, tv.y); one(zv.z, tv.z); one(zv.w, tv.w);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.HW; i += (size_t)gridDim.x * blockDim.x) one(z[i], t[i]);
  }
  sI = block_sum<double>(sI, dred); sS = block_sum<double>(sS, dred); sT = block_sum<double>(sT, dred);
  if (want_bce) sB = block_sum<double>(sB, dred);
  tp = block_sum<long long>(tp, ired); np = block_sum<long long>(np, ired); nt = block_sum<long long>(nt, ired);
  if (threadIdx.x == 0) {
    // per-image replica of the sums (3872 blocks adding to the same three doubles serialised in the L2: 0.15 ms)
    double* rep = a.sums + (size_t)(1 + b) * a.C * DICE_NS;
    atomicAdd(rep + c * DICE_NS + 0, sI); atomicAdd(rep + c * DICE_NS + 1, sS); atomicAdd(rep + c * DICE_NS + 2, sT);
    if (want_bce) atomicAdd(rep + c * DICE_NS + 3, sB);
    if (a.stats) {
      unsigned long long* s = (unsigned long long*)(a.stats + ((size_t)b * a.C + c) * 4);
      atomicAdd(s + 0, (unsigned long long)tp);
      atomicAdd(s + 1, (unsigned long long)(np - tp));
      atomicAdd(s + 2, (unsigned long long)(nt - tp));
    }
  }
}
__global__ void dice_finalize_kernel(const DiceArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double loss = 0.0, bce = 0.0;
    for (int c = 0; c < a.C; ++c) {
      for (int k = 0; k < DICE_NS; ++k) {   // totals = the per-image replicas in image order
        double tot = 0.0;
        for (int b = 0; b < a.B; ++b) tot += a.sums[(size_t)(1 + b) * a.C * DICE_NS + c * DICE_NS + k];
        a.sums[c * DICE_NS + k] = tot;
      }
      const float I = (float)a.sums[c * DICE_NS], S = (float)a.sums[c * DICE_NS + 1], T = (float)a.sums[c * DICE_NS + 2];
      const float score = (2.0f * I) / fmaxf(S, 1e-7f);
      loss += (T > 0.f) ? (double)(1.0f - score) : 0.0;
      bce += a.sums[c * DICE_NS + 3];
    }
    const float dice = (float)(loss / a.C);
    const float bcem = (float)(bce / ((double)a.B * a.C * (double)a.HW));   // reduction='mean' over every element
    *a.loss = a.loss_kind == LOSS_DICE ? dice : a.loss_kind == LOSS_BCE ? bcem : dice + bcem;
    if (a.stats)
      for (int i = 0; i < a.B * a.C; ++i) {
        long long* s = a.stats + (size_t)i * 4;
        s[3] = (long long)a.HW - s[0] - s[1] - s[2];
      }
  }
}
hipError_t launch_dice_fwd(const DiceArgs& a, hipStream_t st) {
  hipError_t e = hipMemsetAsync(a.sums, 0, sizeof(double) * DICE_NS * a.C * (size_t)(1 + a.B), st);   // totals + one replica per image
  if (e != hipSuccess) return e;
  if (a.stats) {
    e = hipMemsetAsync(a.stats, 0, sizeof(long long) * 4 * a.B * a.C, st);
    if (e != hipSuccess) return e;
  }
  if (a.loss_kind < LOSS_DICE || a.loss_kind > LOSS_DICE_BCE) return hipErrorInvalidValue;
  // 32 elements per thread: a workgroup ends in six block reductions and six atomics, which dominated at 8 per thread
  int chunks = (int)((a.HW + 256 * 32 - 1) / (256 * 32));
  if (chunks > 128) chunks = 128;
  if (deterministic_mode()) chunks = 1;   // one workgroup per (image, class): its three sums meet zeroed replicas, no ordering left to chance
  hipLaunchKernelGGL(dice_fwd_kernel, dim3(chunks, a.C, a.B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

// dL/dz for every pixel, written as NHWC rows of CP channels (zero beyond C) so that the head's
// dgrad / wgrad run on the generic conv kernels.
template <typename T>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const DiceArgs a, float grad_scale, void* dl, int CP) {
  constexpr int VEC = EV<T>::VEC;
  constexpr int MAXC = 16;                       // classes kept in registers (launch_dice_bwd checks C <= CP <= 16)
  constexpr int MAXV = MAXC / VEC;
  const size_t npix = (size_t)a.B * a.HW;
  const bool want_dice = a.loss_kind != LOSS_BCE, want_bce = a.loss_kind != LOSS_DICE;
  const float bce_scale = grad_scale / ((float)a.B * (float)a.C * (float)a.HW);
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const size_t b = p / a.HW, i = p - b * a.HW;
    float d[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      d[c] = 0.f;
      if (c < a.C) {
        const float I = (float)a.sums[c * DICE_NS], S = (float)a.sums[c * DICE_NS + 1], Tt = (float)a.sums[c * DICE_NS + 2];
        if (want_bce) {   // d/dz mean(bce_with_logits) = (sigmoid(z) - t) / numel
          const float z = a.logits[(b * a.C + c) * a.HW + i], t = a.target[(b * a.C + c) * a.HW + i];
          d[c] = (sigmoid_acc(z) - t) * bce_scale;
        }
        if (want_dice && Tt > 0.f) {
          const float z = a.logits[(b * a.C + c) * a.HW + i], t = a.target[(b * a.C + c) * a.HW + i];
          // dp/dz = p (1 - p) = e / (1 + e)^2 with e = exp(-|z|): autograd of logsigmoid(z).exp() gives exactly this
          // product (p * e/(1+e) for z >= 0, p * 1/(1+e) for z < 0).  Written as p * (1 - p) it vanishes for z > ~17
          // (p rounds to 1), which drops the gradient of every confidently-positive pixel of a saturated net.
          const float e = expf(-fabsf(z));
          const float dpdz = e / ((1.0f + e) * (1.0f + e));
          float dscore;  // d(2I / max(S, eps)) / dp
          if (S > 1e-7f) dscore = (2.0f * t * S - 2.0f * I) / (S * S);
          else dscore = 2.0f * t / 1e-7f;
          d[c] += -dscore * dpdz * grad_scale / (float)a.C;
        }
      }
    }
    // one NHWC row of CP channels per pixel, written as whole 16-byte vectors (2-byte stores took 0.30 ms per step)
    const size_t row = p * (size_t)(CP / VEC);
#pragma unroll
    for (int v = 0; v < MAXV; ++v)
      if (v < CP / VEC) stv<T>(dl, row + v, EV<T>::pack(d + v * VEC));
  }
}
hipError_t launch_dice_bwd(int dtype, const DiceArgs& a, float grad_scale, void* dlogits, int CP, hipStream_t st) {
  OCTSEG_NO_F16(dtype);
  if (a.C > CP || CP > 16 || CP % 8 != 0) return hipErrorInvalidValue;
  const size_t npix = (size_t)a.B * a.HW;
  const int gr = grid_for(npix, 256);
  if (dtype == DT_F32) hipLaunchKernelGGL(dice_bwd_kernel<float>, dim3(gr), dim3(256), 0, st, a, grad_scale, dlogits, CP);
  else hipLaunchKernelGGL(dice_bwd_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, a, grad_scale, dlogits, CP);
  return hipGetLastError();
}

// ------------------------------------------------------------------ weight packing
// One thread per 16-byte vector of the image.  image[wtap][chunk][ntile][row][q] holds the logical
// chunk (q ^ swz(row)) of output row ntile*BN+row, K range chunk*KC.. (see conv_mfma.hip: the same
// XOR is applied when the MFMA B fragments are read, so the slab can be DMA-copied linearly).
template <typename T>
__global__ __launch_bounds__(256) void pack_image_kernel(const float* w, void* img, int taps, int O, int I, int transpose,
                                                         int BN, int RB, int nchunks, int ntiles) {
  constexpr int VEC = EV<T>::VEC;
  const int VPR = RB / 16, KC = RB / (int)sizeof(T), swz_div = 256 / RB;
  const int rows = transpose ? I : O, K = transpose ? O : I;
  const size_t total = (size_t)taps * nchunks * ntiles * BN * VPR;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
    size_t rem = v;
    const int q = (int)(rem % VPR); rem /= VPR;
    const int row = (int)(rem % BN); rem /= BN;
    const int nt = (int)(rem % ntiles); rem /= ntiles;
    const int chunk = (int)(rem % nchunks);
    const int tap = (int)(rem / nchunks);
    const int lq = q ^ ((row / swz_div) & (VPR - 1));
    const int co = nt * BN + row, c0 = chunk * KC + lq * VEC;
    float x[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int c = c0 + i;
      float val = 0.f;
      if (co < rows && c < K) val = transpose ? w[((size_t)tap * O + c) * I + co] : w[((size_t)tap * O + co) * I + c];
      x[i] = val;
    }
    stv<T>(img, v, EV<T>::pack(x));
  }
}
hipError_t launch_pack_weight_image(int dtype, const float* w, void* img, int taps, int O, int I, int transpose,
                                    const ConvPackInfo& p, hipStream_t st) {
  const size_t total = (size_t)taps * p.nchunks * p.ntiles * p.BN * (p.RB / 16);
  const int gr = grid_for(total, 256, 4096);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(pack_image_kernel<f16_t>, dim3(gr), dim3(256), 0, st, w, img, taps, O, I, transpose, p.BN, p.RB, p.nchunks, p.ntiles);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(pack_image_kernel<float>, dim3(gr), dim3(256), 0, st, w, img, taps, O, I, transpose, p.BN, p.RB, p.nchunks, p.ntiles);
  else
    hipLaunchKernelGGL(pack_image_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, w, img, taps, O, I, transpose, p.BN, p.RB, p.nchunks, p.ntiles);
  return hipGetLastError();
}

// All weight images of a plan in ONE launch: `tab` lists the jobs, `prefix[j]` = first global vector of job j.
template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const float* params, char* ws, const PackJob* tab, const unsigned long long* prefix,
                                                       int njobs, unsigned long long total, int fold) {
  constexpr int VEC = EV<T>::VEC;
  for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs - 1;   // last job whose prefix <= g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const PackJob jb = tab[lo];
    if (fold && jb.src_I != 0) continue;   // images of the tied decomposition: training only, never read by an eval forward
    const float* w = params + jb.src_off;
    const int VPR = jb.RB / 16, KC = jb.RB / (int)sizeof(T), swz_div = 256 / jb.RB;
    const int rows = jb.transpose ? jb.I : jb.O, K = jb.transpose ? jb.O : jb.I, sI = jb.src_I ? jb.src_I : jb.I;
    size_t rem = (size_t)(g - prefix[lo]);
    const size_t v = rem;
    const int q = (int)(rem % VPR); rem /= VPR;
    const int row = (int)(rem % jb.BN); rem /= jb.BN;
    const int nt = (int)(rem % jb.ntiles); rem /= jb.ntiles;
    const int chunk = (int)(rem % jb.nchunks);
    const int tap = (int)(rem / jb.nchunks);
    const int lq = q ^ ((row / swz_div) & (VPR - 1));
    const int co = nt * jb.BN + row, c0 = chunk * KC + lq * VEC;
    // eval: the BatchNorm behind the conv is folded into its forward image, W'[o][i] = W[o][i] * scale[o] (the shift becomes the
    // epilogue's bias): no lazy affine in any consumer's staging
    float osc = 1.f;
    if (fold && !jb.transpose && jb.scale_off != ~(size_t)0 && co < rows) osc = ((const float*)(ws + jb.scale_off))[co];
    float x[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int c = c0 + i;
      float val = 0.f;
      if (jb.tied == 2) {
        // data-gradient image of the masked parity-plane launch: rows = input channels, K = 4 O virtual channels (plane p = c / O), nine
        // tap offsets: plane (py, px) at offset (dy, dx) carries kernel tap r = py + 1 + 2 dy, s = px + 1 + 2 dx where that exists
        if (co < jb.I && c < 4 * jb.O) {
          const int p = c / jb.O, o = c - p * jb.O;
          const int u = (p >> 1) + 1 + 2 * (tap / 3 - 1), vv = (p & 1) + 1 + 2 * (tap % 3 - 1);
          if (u >= 0 && u <= 3 && vv >= 0 && vv <= 3)
            for (int r3 = max(0, 2 - u); r3 <= min(2, 3 - u); ++r3)
              for (int s3 = max(0, 2 - vv); s3 <= min(2, 3 - vv); ++s3) val += w[((size_t)(r3 * 3 + s3) * jb.O + o) * sI + co + jb.src_c0];
        }
      } else if (co < rows && c < K) {
        const int o = jb.transpose ? c : co, ci = (jb.transpose ? co : c) + jb.src_c0;
        if (!jb.tied) val = w[((size_t)tap * jb.O + o) * sI + ci];
        else {   // 4x4 stride-2 kernel of (nearest x2, then the 3x3 source): the source taps that land on the same low-resolution pixel, summed in f32
          const int u = tap >> 2, vv = tap & 3;
          for (int r3 = max(0, 2 - u); r3 <= min(2, 3 - u); ++r3)
            for (int s3 = max(0, 2 - vv); s3 <= min(2, 3 - vv); ++s3) val += w[((size_t)(r3 * 3 + s3) * jb.O + o) * sI + ci];
        }
        if (!jb.transpose) val *= osc;
      }
      x[i] = val;
    }
    stv<T>(ws + jb.dst_off, v, EV<T>::pack(x));
  }
}
hipError_t launch_pack_all(int dtype, const float* params, void* ws, const PackJob* tab, const unsigned long long* prefix, int njobs,
                           unsigned long long total, int fold, hipStream_t st) {
  const int gr = grid_for((size_t)total, 256, 8192);
  if (dtype == DT_F16) hipLaunchKernelGGL(pack_all_kernel<f16_t>, dim3(gr), dim3(256), 0, st, params, (char*)ws, tab, prefix, njobs, total, fold);
  else if (dtype == DT_F32) hipLaunchKernelGGL(pack_all_kernel<float>, dim3(gr), dim3(256), 0, st, params, (char*)ws, tab, prefix, njobs, total, fold);
  else hipLaunchKernelGGL(pack_all_kernel<bf16_t>, dim3(gr), dim3(256), 0, st, params, (char*)ws, tab, prefix, njobs, total, fold);
  return hipGetLastError();
}

// gradient of a tied 4x4 image (and of the skip slice's own 3x3 image) folded back into the 3x3 weight gradient
__global__ __launch_bounds__(256) void tied_fold_kernel(const float* dK4, const float* dW3s, float* dW3, int O, int Ca, int Cs) {
  const int I = Ca + Cs;
  const size_t total = (size_t)9 * O * I;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const int o = (int)((idx / I) % O);
    const int t = (int)(idx / ((size_t)I * O));
    float val;
    if (i < Ca) {
      const int r3 = t / 3, s3 = t - 3 * r3;   // tap r3 lies in A(2 - r3) and A(3 - r3)
      val = 0.f;
      for (int u = 2 - r3; u <= 3 - r3; ++u)
        for (int v = 2 - s3; v <= 3 - s3; ++v) val += dK4[((size_t)(u * 4 + v) * O + o) * Ca + i];
    } else {
      val = dW3s[((size_t)t * O + o) * Cs + (i - Ca)];
    }
    dW3[idx] += val;
  }
}
hipError_t launch_tied_fold(const float* dK4, const float* dW3s, float* dW3, int O, int Ca, int Cs, hipStream_t st) {
  hipLaunchKernelGGL(tied_fold_kernel, dim3(grid_for((size_t)9 * O * (Ca + Cs), 256)), dim3(256), 0, st, dK4, dW3s, dW3, O, Ca, Cs);
  return hipGetLastError();
}

// ------------------------------------------------------------------ serving: threshold + nearest resize + mask assembly
// out[n][y][x][out_ch] = logits[n][ch][rows[y]][cols[x]] > 0  (sigmoid(z) > 0.5).  rows / cols: source index of every
// output row / column (the caller builds them with the resize rule it wants bit-for-bit: the host mirror passes OpenCV's
// INTER_NEAREST table, predict.py:92-96); null tables = floor((i + 0.5) * S / O) in exact integers (identity when S == O).
__global__ __launch_bounds__(256) void mask_assemble_kernel(const float* logits, int N, int C, int SH, int SW, int ch, float* out, int OH,
                                                            int OW, int OC, int out_ch, const int* rows, const int* cols) {
  const size_t total = (size_t)N * OH * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW);
    const int y = (int)((i / OW) % OH);
    const int n = (int)(i / ((size_t)OW * OH));
    int sy = rows ? rows[y] : (int)(((long long)(2 * y + 1) * SH) / (2 * OH));
    int sx = cols ? cols[x] : (int)(((long long)(2 * x + 1) * SW) / (2 * OW));
    sy = min(max(sy, 0), SH - 1); sx = min(max(sx, 0), SW - 1);
    const float z = logits[(((size_t)n * C + ch) * SH + sy) * SW + sx];
    out[i * OC + out_ch] = sigmoid_acc(z) > 0.5f ? 1.f : 0.f;   // the reference's `sigmoid() > 0.5` in fp32 (model.py:195), as the Dice kernel
  }
}
hipError_t launch_mask_assemble(const float* logits, int N, int C, int SH, int SW, int ch, float* out, int OH, int OW, int OC, int out_ch,
                                const int* rows, const int* cols, hipStream_t st) {
  hipLaunchKernelGGL(mask_assemble_kernel, dim3(grid_for((size_t)N * OH * OW, 256)), dim3(256), 0, st, logits, N, C, SH, SW, ch, out, OH, OW,
                     OC, out_ch, rows, cols);
  return hipGetLastError();
}

// ------------------------------------------------------------------ fused optimizers (torch defaults)
__global__ __launch_bounds__(256) void optim_kernel(const OptArgs a, float bc1, float bc2, float radam_rect, int radam_use) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float p = a.p[i];
    float g = a.g[i] * a.grad_scale;
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    if (a.kind == 0) {  // SGD
      p -= a.lr * g;
    } else if (a.kind == 2) {  // RMSprop
      float v = a.alpha * a.v[i] + (1.f - a.alpha) * g * g;
      a.v[i] = v;
      p -= a.lr * g / (sqrtf(v) + a.eps);
    } else {
      float m = a.m[i] + (g - a.m[i]) * (1.f - a.beta1);  // lerp, as torch does
      float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
      a.m[i] = m; a.v[i] = v;
      if (a.kind == 1) {  // Adam
        const float denom = sqrtf(v) / sqrtf(bc2) + a.eps;
        p -= (a.lr / bc1) * (m / denom);
      } else {  // RAdam
        const float mhat = m / bc1;
        if (radam_use) p -= mhat * a.lr * (sqrtf(bc2) / (sqrtf(v) + a.eps)) * radam_rect;
        else p -= mhat * a.lr;
      }
    }
    a.p[i] = p;
  }
}
hipError_t launch_optim_step(const OptArgs& a, hipStream_t st) {
  const double b1t = pow((double)a.beta1, (double)a.step), b2t = pow((double)a.beta2, (double)a.step);
  const float bc1 = (float)(1.0 - b1t), bc2 = (float)(1.0 - b2t);
  float rect = 1.f; int use = 0;
  if (a.kind == 3) {
    const double rho_inf = 2.0 / (1.0 - (double)a.beta2) - 1.0;
    const double rho_t = rho_inf - 2.0 * a.step * b2t / (1.0 - b2t);
    if (rho_t > 5.0) {
      use = 1;
      rect = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
    }
  }
  hipLaunchKernelGGL(optim_kernel, dim3(grid_for(a.n, 256)), dim3(256), 0, st, a, bc1, bc2, rect, use);
  return hipGetLastError();
}

}  // namespace octseg
